#!/usr/bin/env python3
"""bench.py - board-steps/sec of the vectorized 2048 environment on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1]
(C2) = M=65536 boards x N=128 random-action env steps, in bit-exact ("table") spawn mode, on
each GPU.  Boards are independent games, so N GPUs run N independent shards (weak scaling, no
data-path collective; the only collective is the timing barrier / max).  `value` is the whole-job
board-steps/s with all inputs resident in HBM.

The JSON line also carries
  roofline      the dominant kernel's algorithmic bytes / its HIP-event-timed duration vs 8 TB/s
  cpu_baseline  the CPU oracle (a bit-exact C/OpenMP restatement of the reference, "port") timed
                on this host on a bounded sample of the same workload (rank 0, N=1 only)
  stepwise      the same workload through the per-step API (prepare/sample/step launches)
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_BOARD_STEP = 99  # SURVEY.md 8d / BASELINE.md 4: algorithmic bytes of one env step (no obs)
BYTES_PER_BOARD_STEP_OBS = 1128  # with masked sampling + fp32 one-hot
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)  # 0.12 s timed at C2: long enough to average out host jitter
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=65536, help="M per GPU (C2: 65536)")
    ap.add_argument("--env-steps", type=int, default=128, help="N env steps per bench step (C2: 128)")
    ap.add_argument("--mode", choices=["fused", "fused-traj", "stepwise", "obs"], default="fused",
                    help="fused: one rollout launch per bench step (default); stepwise: per-step API; "
                         "obs: per-step API + masked sampling + one-hot (C3 pipeline)")
    ap.add_argument("--masked", type=int, default=0, help="masked-uniform actions (default: unmasked, C2)")
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--step-mapping", choices=["lane", "wave"], default="lane",
                    help="per-step kernel mapping: lane-per-board (default) or one wavefront per board")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary stepwise measurement")
    ap.add_argument("--ids", choices=["deferred", "inline"], default="deferred",
                    help="fused mode: where the ordered game-id scan of a rollout runs - 'deferred' = on a side "
                         "stream beside the next rollout's kernel, 'inline' = on the launch stream")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket every k-th timed launch of the dominant kernel with HIP events (the pair of "
                         "timestamp packets costs a few us of GPU idle per launch, so not every launch carries one)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-over-ranks (nccl = RCCL; gloo only for "
                         "rehearsing the N>1 launch contract on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: map every rank to cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--schedule", choices=["auto", "host", "host-async", "device"], default="auto",
                    help="where the bit-exact mode's PCG64 draw schedule is computed: host (synchronous native call), "
                         "host-async (pipelined native producer threads drawing ahead), device (csrc/devrng2.hip: no host "
                         "threads); auto (default) = host-async when this rank's CPU share allows at least 6 producer "
                         "workers, else device")
    ap.add_argument("--prefetch", type=int, default=0,
                    help="draw the next rollout's host schedule on a worker thread (default 0: the async kernel "
                         "launch already overlaps the host schedule of rollout k+1 with the GPU running rollout k)")
    return ap.parse_args()


def host_cpu_share() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(M: int, N: int, seed: int, seconds: float):
    """Time the CPU oracle on the host cores: repeated N-step rollouts of M boards (same seeds,
    same action stream definition) until ~`seconds` have elapsed."""
    from oracle import oracle as orc

    lib_path = None
    try:  # a -march=native build for a fair host number (the in-tree .so is generic x86-64-v2)
        tmp = tempfile.mkdtemp(prefix="ml2048_orc_")
        lib_path = os.path.join(tmp, "libml2048_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o",
                               lib_path, os.path.join(ROOT, "oracle", "ml2048_oracle.c"), "-lm"],
                              stderr=subprocess.DEVNULL)
        orc._lib = orc.lib(lib_path)
    except Exception:
        orc.build()
    threads = host_cpu_share()
    orc.lib().orc_set_num_threads(threads)
    env = orc.OracleVecGame(M, "normal")
    env.reset(seed)
    counter = 0

    def one_rollout():
        nonlocal counter
        for _ in range(N):
            env.prepare()
            a, _ = orc.sample_actions(env._data["valid_actions"], 7, counter, masked=False)
            counter += 1
            env.step(a)

    one_rollout()  # warm-up (deals every game, touches memory)
    t0 = time.perf_counter()
    reps = 0
    while True:
        one_rollout()
        reps += 1
        if time.perf_counter() - t0 >= seconds * 0.75 or reps >= 10000:
            break
    dt = time.perf_counter() - t0
    # single-thread figure on a smaller slice of the same workload
    orc.lib().orc_set_num_threads(1)
    t1 = time.perf_counter()
    reps1 = 0
    while True:
        one_rollout()
        reps1 += 1
        if time.perf_counter() - t1 >= seconds * 0.25 or reps1 >= 10000:
            break
    dt1 = time.perf_counter() - t1
    orc.lib().orc_set_num_threads(threads)
    try:
        cpu = subprocess.check_output("lscpu | grep 'Model name' | head -1", shell=True, text=True).split(":")[-1].strip()
    except Exception:
        cpu = "unknown"
    return {
        "value": M * N * reps / dt, "unit": "board-steps/s", "cores": threads, "kind": "port",
        "value_1thread": M * N * reps1 / dt1,
        "sample": f"{reps} x (M={M} boards x N={N} steps) in {dt:.1f}s: C/OpenMP restatement of game_numba "
                  f"(bit-exact oracle, not numba; native prepare loop; Python driver loop per env step as in the "
                  f"reference), {threads} OpenMP threads = this box's CPU share of {os.cpu_count()} logical CPUs, {cpu}",
    }


def main():
    args = parse()
    import torch

    import ml2048_amd as amd
    from ml2048_amd import _lib

    L = _lib.load()
    from ml2048_amd import build as hip_build

    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        hip_build.build()  # no-op when the in-tree library is up to date
    else:  # other ranks never build; they wait for rank 0 if it had to
        t_wait = time.time()
        while hip_build.is_stale() and time.time() - t_wait < 300:
            time.sleep(0.5)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend="gloo")
        dist = dist_mod

    # host threads of the schedule producer: this rank's share of the box's CPUs (cgroup quota / ranks
    # on this node), minus the Python thread, the stitcher and the HIP runtime's own threads; 8 workers draw a
    # 128-step schedule in ~0.09 ms, which keeps the stream ahead of the ~0.11 ms rollout kernel
    if "ML2048_HOST_THREADS" not in os.environ:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        share = max(1, host_cpu_share() // max(1, local_world))
        os.environ["ML2048_HOST_THREADS"] = str(max(1, min(8, share - 4)))
    if args.schedule == "auto":
        # the host stream needs ~6 workers to stay ahead of the 0.09 ms rollout kernel (4 workers: 0.149 ms per
        # rollout; measured); on a rank with fewer CPUs the schedule drawn on the GPU (0.128 ms, no host threads) wins
        args.schedule = "host-async" if int(os.environ["ML2048_HOST_THREADS"]) >= 6 else "device"

    M, N, K, W = args.boards, args.env_steps, args.steps, args.warmup
    # shard `rank` of a 2-level game: independent replica with its own seed (SURVEY 8e replica mode)
    env = amd.VecGame(M, amd.reward_fn_normal, device=dev, idx_base=0, step_mapping=args.step_mapping,
                      schedule=args.schedule)
    env.reset(args.seed + rank)
    pol = amd.RandomPolicy(seed=7 + rank, masked=bool(args.masked))
    obs = torch.empty((M, 16, 4, 4), dtype=torch.float32, device=dev) if args.mode == "obs" else None
    counter = [0]
    kern_ms = []

    traj = None
    if args.mode == "fused-traj":  # every step's transition materialised: 54 B written per board-step
        from ml2048_amd import train_ops

        traj = train_ops.new_buffers((N, M), dev, spec=train_ops.REPLAY_SPEC)

    def bench_step(timed: bool):
        if args.mode in ("fused", "fused-traj"):
            tm = None
            if timed:  # HIP event pair attached to the kernel's dispatch (hipExtLaunchKernel), on its stream
                import ctypes as C
                tm = C.c_void_p()
                _lib.check(L.ml2048_timer_create(C.byref(tm)), "timer_create")
            env.rollout(N, masked=bool(args.masked), act_seed=7 + rank, act_counter=counter[0], kernel_timer=tm,
                        prefetch_next=bool(args.prefetch), trajectory=traj,
                        finish_ids="deferred" if args.ids == "deferred" else True)
            counter[0] += N
            if tm is not None:
                kern_ms.append(tm)
        else:
            for _ in range(N):
                env.prepare_torch()
                a, _ = pol.sample_actions_u8(env.valid_actions)
                env.step_torch(a)
                if obs is not None:
                    env.onehot(out=obs)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(W):
        bench_step(False)
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        bench_step(k % max(1, args.event_every) == 0)
    barrier()
    dt = time.perf_counter() - t0
    from ml2048_amd.shard import max_over_ranks

    dt = max_over_ranks(dt, dist)

    total_board_steps = world * M * N * K
    value = total_board_steps / dt

    # dominant kernel duration, HIP events on the launch stream
    roofline = None
    if args.mode in ("fused", "fused-traj") and kern_ms:
        import ctypes as C
        vals = []
        for tm in kern_ms:
            ms = C.c_float(0.0)
            _lib.check(L.ml2048_timer_elapsed_ms(tm, C.byref(ms)), "timer_elapsed_ms")
            vals.append(float(ms.value))
            L.ml2048_timer_destroy(tm)
        avg_ms = sum(vals) / len(vals)
        # fused-traj: the 54 B/board-step of REPLAY_SPEC transitions are really written
        alg_bytes = (54 if args.mode == "fused-traj" else BYTES_PER_BOARD_STEP) * M * N
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_rollout", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "note": ("algorithmic = 99 B/board-step x M x N per launch (what N per-step launches must move); "
                             "the fused kernel keeps boards in registers, so measured HBM traffic is far below it")
                    if args.mode == "fused" else
                    "algorithmic = the 54 B/board-step of REPLAY_SPEC transitions this launch really writes"}
    else:
        # per-step API: time the step kernel alone over a few launches
        evs = []
        a, _ = pol.sample_actions_u8(env.valid_actions)
        for _ in range(20):
            env.prepare_torch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.step_torch(a)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize(dev)
        avg_ms = sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2]
        per = BYTES_PER_BOARD_STEP
        achieved = per * M / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_step" if args.step_mapping == "lane" else "k_step_wave", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": per * M}
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if roofline and os.path.exists(traffic_file):
        try:
            tj = json.load(open(traffic_file))
            key = f"{roofline['kernel']}:M={M}:N={N if args.mode == 'fused' else 1}" + (":traj" if args.mode == "fused-traj" else "")
            if key in tj:
                roofline["traffic"] = tj[key]["bytes_per_launch"]
                roofline["traffic_source"] = tj[key].get("source")
        except Exception:
            pass

    extras = {}
    if rank == 0 and world == 1 and args.mode == "fused" and not args.no_extras:
        # secondary: the same workload through the per-step API (what an RL loop with a policy uses)
        env2 = amd.VecGame(M, amd.reward_fn_normal, device=dev)
        env2.reset(args.seed)
        pol2 = amd.RandomPolicy(seed=7, masked=bool(args.masked))

        def stepwise():
            for _ in range(N):
                env2.prepare_torch()
                a, _ = pol2.sample_actions_u8(env2.valid_actions)
                env2.step_torch(a)

        stepwise()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            stepwise()
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t1
        extras["stepwise"] = {"value": M * N * reps / d1, "unit": "board-steps/s",
                              "us_per_env_step": d1 / (N * reps) * 1e6,
                              "note": "per-step API: 3 kernel launches per env step (prepare, sample, step); host-launch-bound"}

    if rank == 0 and world == 1 and args.mode == "fused" and not args.no_extras:
        # secondary: the same fused workload in rocRAND spawn mode (no host PCG64 schedule; statistically the same
        # game, bit-exact with the oracle's counter-based spawn, NOT with numba) - purely GPU-bound
        env3 = amd.VecGame(M, amd.reward_fn_normal, device=dev, spawn="rocrand", spawn_seed=args.seed)
        env3.reset(args.seed)
        for _ in range(3):
            env3.rollout(N, masked=bool(args.masked), act_seed=7)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        reps = 50
        for r in range(reps):
            env3.rollout(N, masked=bool(args.masked), act_seed=7, act_counter=(r + 3) * N)
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t1
        extras["rocrand_spawn_mode"] = {"value": M * N * reps / d1, "unit": "board-steps/s", "ms_per_step": d1 / reps * 1e3,
                                        "note": "spawn='rocrand': fused rollout with the rocRAND Philox spawn, no host "
                                                "schedule; not bit-exact with the numba path (see DESIGN.md)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(M, N, args.seed, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "env board-steps/sec (M boards x N steps), bit-exact vs game_numba semantics",
            "value": value, "unit": "board-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"C2: M={M} boards/GPU x N={N} random-action env steps per bench step "
                                   f"({'masked' if args.masked else 'unmasked'} uniform actions sampled on device, "
                                   f"table spawn = bit-exact mode), mode={args.mode}",
                       "boards_per_gpu": M, "env_steps_per_step": N, "mode": args.mode, "sharding": f"replicas x{world}",
                       "host_schedule": args.schedule, "id_scan": args.ids, "spawn": "table (bit-exact with the reference's PCG64 tables)"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extras)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
