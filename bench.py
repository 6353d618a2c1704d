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
  roofline      for the dominant kernel, against the bound it actually touches:
                  fused modes     VALU issue: wave-instructions/s (instructions per wave-step from a committed
                                  rocprofv3 SQ-counter profile x waves x steps / the kernel's HIP-event duration)
                                  vs CUs x 4 SIMDs x clock / 2 cycles; the 99 B/board-step HBM-equivalent is a
                                  side figure, never `frac`
                  per-step modes  HBM: algorithmic bytes of the one launch per env step / its duration vs 8 TB/s
  cpu_baseline  the CPU oracle (a bit-exact C/OpenMP restatement of the reference, "port") timed
                on this host on a bounded sample of the same workload (rank 0, N=1 only)
  extras        (rank 0, N=1, default mode) the same C2 workload one launch per env step, C3 (M=262144, masked
                sampling + one-hot observation per env step), the fused rollout at C4's per-GPU shard size, the
                rocRAND spawn mode, and the C2 rollout with actions drawn over the valid actions only
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
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
# VALU issue (same guide, "SIMD-32": a wave64 VALU instruction issues over 2 cycles; one wave ALONE on its SIMD
# sustains one per 4): 256 CUs x 4 SIMDs x 2.4 GHz / 2
N_SIMD, CLOCK_GHZ = 256 * 4, 2.4
VALU_PEAK_GINST = N_SIMD * CLOCK_GHZ / 2.0
LONE_WAVE_PEAK_GINST = N_SIMD * CLOCK_GHZ / 4.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)  # 0.1 s timed at C2: long enough to average out host jitter
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=65536, help="M per GPU (C2: 65536)")
    ap.add_argument("--env-steps", type=int, default=128, help="N env steps per bench step (C2: 128)")
    ap.add_argument("--mode", choices=["fused", "fused-traj", "stepwise", "obs", "stepwise3"], default="fused",
                    help="fused: one rollout launch per bench step (default); stepwise: one ml2048_step_prepare "
                         "launch per env step; obs: the same with masked sampling + the one-hot observation "
                         "(C3 pipeline); stepwise3: the reference-shaped per-step API (prepare, sample, step launches)")
    ap.add_argument("--masked", type=int, default=None, help="masked-uniform actions (default: unmasked, C2; obs: masked)")
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--spawn", choices=["table", "rocrand"], default="table",
                    help="table (default): the reference's PCG64 tables, bit-exact with numba; rocrand: the rocRAND Philox "
                         "spawn (same game statistically, no host schedule)")
    ap.add_argument("--step-mapping", choices=["lane", "wave"], default="lane",
                    help="stepwise3: lane-per-board step kernel (default) or one wavefront per board")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--ids", choices=["deferred", "inline"], default="deferred",
                    help="fused mode: where the ordered game-id scan of a rollout runs - 'deferred' = on a side "
                         "stream beside the next rollout's kernel, 'inline' = on the launch stream")
    ap.add_argument("--event-every", type=int, default=0,
                    help="attach a HIP event pair to every k-th timed launch of the dominant kernel (0 = auto: about 32 "
                         "samples per measurement, never more often than every 8th launch - a timed launch costs ~8 us "
                         "end to end, tools/debug/event_ab.sh)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-over-ranks (nccl = RCCL; gloo only for "
                         "rehearsing the N>1 launch contract on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: map every rank to cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--schedule", choices=["auto", "host", "host-async", "device"], default="auto",
                    help="where the bit-exact mode's PCG64 draw schedule is computed: host (synchronous native call), "
                         "host-async (pipelined native producer threads drawing ahead), device (csrc/devrng2.hip: no host "
                         "threads; opt-in); auto (default) = host-async sized to this rank's CPU share "
                         "(ml2048_amd.shard.host_thread_plan)")
    ap.add_argument("--min-warmup-ms", type=float, default=30.0,
                    help="the W warm-up steps are repeated until at least this much time has passed: a bench step here is "
                         "~0.07 ms, and the GPU needs ~3 ms of work after an idle stretch to reach its loaded clocks (with "
                         "W = 5 the timed region ran 5 %% slower kernels than a long run: tools/debug/warm_ab.sh); "
                         "0 = exactly W steps.  The line reports the count as warmup_steps_run")
    ap.add_argument("--prefetch", type=int, default=0,
                    help="draw the next rollout's host schedule on a worker thread (synchronous host schedule only)")
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


# ---- roofline bookkeeping ------------------------------------------------------------------------------------------
def _inst_table():
    """profiles/valu_insts.json: instructions per wave-step of the fused rollout kernel by (variant, pass), each
    entry naming the committed rocprofv3 SQ-counter summary it was taken from."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "valu_insts.json")))
    except Exception:
        return {}


def _json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def _traffic(key: str):
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if key in tj:
            return tj[key]["bytes_per_launch"], tj[key].get("source")
    except Exception:
        pass
    return None, None


def rollout_roofline(M: int, N: int, avg_ms: float, masked: bool, traj: bool, spawn: str = "table"):
    """VALU-issue roofline of k_rollout (fused) / HBM-write roofline of the trajectory-writing instantiation."""
    waves = (M + 63) // 64
    if traj:
        alg = 54 * M * N  # the REPLAY_SPEC transitions this launch really writes
        ach = alg / (avg_ms * 1e-3) / 1e9
        tb, src = _traffic(f"k_rollout:M={M}:N={N}:traj")
        return {"bound": "hbm", "kernel": "k_rollout<EXTRAS> (trajectory)", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tb, "traffic_source": src,
                "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg,
                "note": "algorithmic = the 54 B/board-step of REPLAY_SPEC transitions this launch writes"}
    variant = ("masked" if masked else "plain") + ("" if spawn == "table" else "-rocrand")
    pass_ = "small" if M <= int(os.environ.get("ML2048_PLAIN_ILP_MAX", "65536")) else "large"  # launch_rollout's switch
    ent = _inst_table().get(f"k_rollout:{variant}:{pass_}")
    tb, src = _traffic(f"k_rollout:M={M}:N={N}")
    r = {"bound": "valu_issue", "kernel": f"k_rollout<{variant}, {pass_}-M pass>", "unit": "G wave-instructions/s",
         "peak": VALU_PEAK_GINST, "avg_launch_ms": avg_ms, "traffic": tb, "traffic_source": src,
         "peak_note": "256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)"}
    if ent:
        valu = ent["valu_per_wave_step"]
        ach = valu * waves * N / (avg_ms * 1e-3) / 1e9
        r.update({"achieved": ach, "frac": ach / VALU_PEAK_GINST, "insts_per_wave_step": valu,
                  "insts_source": ent["source"], "insts_measured_at": ent.get("measured_at")})
        costs = _json("valu_issue_costs.json")  # tools/microbench/valu_rate.hip on this chip
        if ent.get("all_per_wave_step") and waves <= N_SIMD and costs.get("lone_wave_cycles_per_inst"):
            # one wave per SIMD (or fewer): every instruction of the lone wave, of any type, takes one issue slot - measured
            # at ~5 cycles per instruction (the guide's 4 is the FP32 FMA's; v_cmp and v_readlane take ~8.5).  The share
            # of the kernel's time that the wave spends issuing at that cadence:
            cyc = costs["lone_wave_cycles_per_inst"]
            r["lone_wave_frac"] = ent["all_per_wave_step"] * waves * N / (avg_ms * 1e-3) / 1e9 / \
                (min(waves, N_SIMD) * CLOCK_GHZ / cyc)
            r["lone_wave_cycles_per_inst"] = cyc
            r["lone_wave_note"] = ("all instructions per wave-step x the measured issue cadence of a wave alone on its SIMD "
                                   f"({cyc} cycles, {costs.get('source')}) / the step's duration at {CLOCK_GHZ} GHz")
        mix = _json("valu_mix.json").get(f"k_rollout:{variant}:{pass_}")
        if mix and waves > N_SIMD and costs.get("busy_simd_ns_full_rate"):
            # several waves per SIMD: the SIMD's vector pipe is the shared resource; an instruction of this loop costs it
            # mean_cost_vs_full_rate x the measured full-rate cost (half of the mix - v_perm_b32, v_cndmask_b32, v_dot4,
            # three-operand VOP3 - issues at half rate).  Share of the step that pipe is busy:
            per_inst_ns = mix["mean_cost_vs_full_rate"] * costs["busy_simd_ns_full_rate"]
            r["valu_pipe_busy_frac"] = valu * (waves / N_SIMD) * per_inst_ns * 1e-9 / (avg_ms * 1e-3 / N)
            r["valu_pipe_note"] = (f"{valu} vector instructions per wave-step x {waves / N_SIMD:.2f} waves per SIMD x "
                                   f"{per_inst_ns:.3f} ns per instruction of this loop's mix on a busy SIMD "
                                   f"(profiles/valu_mix.json, {costs.get('source')}) / the step's duration")
        r["frac_note"] = ("frac = vector instructions issued per second / the chip's 2-cycle VALU peak: it FALLS when work is "
                          "removed (round 2's kernel issued 211 VALU per wave-step in 68 us: 0.31; this one 171 in 65 us), and "
                          "half of the mix issues at half rate on gfx950 (profiles/valu_issue_costs.json) - lone_wave_frac "
                          "(one wave per SIMD) / valu_pipe_busy_frac (several) price the kernel's time at the measured costs")
        assert r["frac"] <= 1.0, r
    else:
        r.update({"achieved": None, "frac": None, "insts_per_wave_step": None,
                  "note": f"no SQ-counter profile committed for k_rollout:{variant}:{pass_} (profiles/valu_insts.json)"})
    eq = BYTES_PER_BOARD_STEP * M * N / (avg_ms * 1e-3) / 1e9
    r["hbm_equivalent_99B"] = {"GB/s": eq, "x_of_8TBs": eq / HBM_PEAK_GBS,
                               "note": "what N per-step launches would have to move for the same work; the fused kernel "
                                       "keeps boards in registers (see traffic), so this is NOT a fraction of any roofline"}
    return r


def step_prepare_roofline(M: int, avg_ms: float, obs: bool, sampler: bool):
    per = (BYTES_PER_BOARD_STEP_OBS if obs else BYTES_PER_BOARD_STEP) + 1  # + the terminated byte the fused kernel reads
    if sampler and not obs:
        per += 5  # next action 1 + log-prob 4 (already part of the 1128)
    alg = per * M
    ach = alg / (avg_ms * 1e-3) / 1e9
    tb, src = _traffic(f"k_step_prepare:M={M}:obs={int(obs)}")
    return {"bound": "hbm", "kernel": "k_step_prepare" + (" (+ one-hot, masked sampler)" if obs else " (+ sampler)"),
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tb,
            "traffic_source": src, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg,
            "bytes_per_board_step": per}


def main():
    args = parse()
    import torch

    import ml2048_amd as amd
    from ml2048_amd import _lib
    from ml2048_amd import build as hip_build
    from ml2048_amd.shard import host_thread_plan, max_over_ranks

    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        hip_build.build()  # no-op when the in-tree library is up to date
    else:  # other ranks never build; they wait for rank 0 if it had to
        t_wait = time.time()
        while hip_build.is_stale() and time.time() - t_wait < 300:
            time.sleep(0.5)
    L = _lib.load()

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

    # host threads of the schedule producer: this rank's share of the node's CPUs (shard.host_thread_plan)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    plan = host_thread_plan(host_cpu_share(), local_world, args.schedule)
    if "ML2048_HOST_THREADS" not in os.environ:
        os.environ["ML2048_HOST_THREADS"] = str(plan["threads"])
    args.schedule = plan["schedule"]
    rank_info = {"rank": rank, "host_schedule": args.schedule, "ML2048_HOST_THREADS": int(os.environ["ML2048_HOST_THREADS"]),
                 "cpus_per_rank": plan["cpus_per_rank"]}

    M, N, K, W = args.boards, args.env_steps, args.steps, args.warmup
    masked = bool(args.masked) if args.masked is not None else args.mode == "obs"

    warm_timers = []  # timers of warm-up launches: kept alive until the run ends
    warm_run = [0]    # warm-up steps the last measure() really ran (W, extended to --min-warmup-ms)

    def new_timer():
        tm = C.c_void_p()
        _lib.check(L.ml2048_timer_create(C.byref(tm)), "timer_create")
        return tm

    def timers_avg_ms(timers):
        vals = []
        for tm in timers:
            ms = C.c_float(0.0)
            _lib.check(L.ml2048_timer_elapsed_ms(tm, C.byref(ms)), "timer_elapsed_ms")
            vals.append(float(ms.value))
            L.ml2048_timer_destroy(tm)
        return sum(vals) / len(vals) if vals else None

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(step_fn, k, w, finish=None):
        """w untimed + k timed bench steps, barrier + synchronize on both sides, max over ranks.  `finish` (fused modes:
        VecGame.join_ids) issues whatever the steps deferred - the ordered game-id scan of the LAST rollout, which no later
        rollout() call is there to issue - so that it runs, and is waited for, inside the region it belongs to."""
        # Everything alive now is set-up that lives as long as the run: collect once and move it out of the cyclic
        # collector's sight.  A full collection with torch imported takes ~40 ms (tools/debug/hiccup_trace.py: one per
        # run, around the 750th rollout) - 500 rollouts' worth of a launch thread that has to stay ahead of a 70 us
        # kernel; after the freeze a collection only walks what the loop itself allocated.
        gc.collect()
        gc.freeze()
        t_warm = time.perf_counter()
        for j in range(w):
            # one warm-up launch carries the kernel timer: whatever the first timed dispatch of a process sets up in the
            # runtime happens here, not inside the timed region (its time is not a sample)
            step_fn("warm" if j == 0 else False)
        warm_run[0] = w
        while w > 0 and (time.perf_counter() - t_warm) * 1e3 < args.min_warmup_ms:  # see --min-warmup-ms
            step_fn(False)
            warm_run[0] += 1
        if finish is not None:
            finish()
        barrier()
        every = args.event_every if args.event_every > 0 else max(8, k // 32)
        t0 = time.perf_counter()
        for i in range(k):
            step_fn(i % every == min(every // 2, k - 1))  # (not the window's first launch: it starts on an idle GPU)
        if finish is not None:
            finish()
        barrier()
        return max_over_ranks(time.perf_counter() - t0, dist)

    # ---- workloads ----------------------------------------------------------------------------------------------
    def make_fused(m, n, masked_, mode, schedule, seed_off=0, spawn="table"):
        env = amd.VecGame(m, amd.reward_fn_normal, device=dev, schedule=schedule, spawn=spawn, spawn_seed=args.seed)
        env.reset(args.seed + rank + seed_off)
        traj = None
        if mode == "fused-traj":  # every step's transition materialised: 54 B written per board-step
            from ml2048_amd import train_ops

            traj = train_ops.new_buffers((n, m), dev, spec=train_ops.REPLAY_SPEC)
        counter = [0]
        timers = []

        def step(timed):
            tm = new_timer() if timed else None
            env.rollout(n, masked=masked_, act_seed=7 + rank, act_counter=counter[0], kernel_timer=tm,
                        prefetch_next=bool(args.prefetch), trajectory=traj,
                        finish_ids="deferred" if args.ids == "deferred" else True)
            counter[0] += n
            if tm is not None:
                (timers if timed is True else warm_timers).append(tm)  # a warm-up launch's time is not a sample

        return step, timers, env

    def make_stepwise(m, n, masked_, obs_, seed_off=0):
        """one ml2048_step_prepare launch per env step; the RandomPolicy draw for the next step comes out of it"""
        env = amd.VecGame(m, amd.reward_fn_normal, device=dev)
        env.reset(args.seed + rank + seed_off)
        env.prepare_torch()
        pol = amd.RandomPolicy(seed=7 + rank, masked=masked_)
        obs_buf = torch.empty((m, 16, 4, 4), dtype=torch.float32, device=dev) if obs_ else None
        cur = [pol.sample_actions_u8(env.valid_actions)[0]]
        timers = []

        def step(timed):
            a = cur[0]
            for j in range(n):
                tm = None
                if timed and j % 16 == 0:
                    tm = new_timer()
                    _lib.check(L.ml2048_timer_arm(tm), "timer_arm")
                    (timers if timed is True else warm_timers).append(tm)
                a = env.step_prepare_torch(a, sample=pol, onehot=obs_buf)["actions"]
            cur[0] = a

        return step, timers, env

    def make_stepwise3(m, n, masked_):
        env = amd.VecGame(m, amd.reward_fn_normal, device=dev, step_mapping=args.step_mapping)
        env.reset(args.seed + rank)
        pol = amd.RandomPolicy(seed=7 + rank, masked=masked_)

        def step(timed):
            for _ in range(n):
                env.prepare_torch()
                a, _ = pol.sample_actions_u8(env.valid_actions)
                env.step_torch(a)

        return step, [], env

    if args.mode in ("fused", "fused-traj"):
        step_fn, timers, env = make_fused(M, N, masked, args.mode, args.schedule, spawn=args.spawn)
    elif args.mode in ("stepwise", "obs"):
        step_fn, timers, env = make_stepwise(M, N, masked, args.mode == "obs")
    else:
        step_fn, timers, env = make_stepwise3(M, N, masked)
    dt = measure(step_fn, K, W, finish=env.join_ids)
    warmup_steps_run = warm_run[0]
    value = world * M * N * K / dt
    avg_ms = timers_avg_ms(timers)

    if args.mode in ("fused", "fused-traj"):
        roofline = rollout_roofline(M, N, avg_ms, masked, args.mode == "fused-traj", args.spawn)
    elif args.mode in ("stepwise", "obs"):
        roofline = step_prepare_roofline(M, avg_ms, args.mode == "obs", True)
    else:
        roofline = None  # three small launches per env step: host-launch-bound, no single dominant kernel to price

    extras = {}
    if rank == 0 and world == 1 and args.mode == "fused" and not args.no_extras:
        saved_every = args.event_every
        # (1) the same C2 workload, one launch per env step (what a policy-driven RL loop uses)
        sfn, stm, _e = make_stepwise(M, N, masked, False, seed_off=100)
        d1 = measure(sfn, 20, 3, finish=_e.join_ids)
        ms1 = timers_avg_ms(stm)
        extras["stepwise"] = {"value": M * N * 20 / d1, "unit": "board-steps/s", "us_per_env_step": d1 / (N * 20) * 1e6,
                              "workload": f"M={M} x N={N}, one ml2048_step_prepare launch per env step (step + re-deal + "
                                          "next RandomPolicy draw), bit-exact spawn mode",
                              "roofline": step_prepare_roofline(M, ms1, False, True)}
        del sfn, _e
        # (2) C3 = BASELINE.json configs[2]: M=262144, masked sampling + one-hot (M,16,4,4) f32 observation per env step
        sfn, stm, _e = make_stepwise(262144, N, True, True, seed_off=200)
        d1 = measure(sfn, 8, 2, finish=_e.join_ids)
        ms1 = timers_avg_ms(stm)
        extras["c3_obs"] = {"value": 262144 * N * 8 / d1, "unit": "board-steps/s", "us_per_env_step": d1 / (N * 8) * 1e6,
                            "workload": f"C3: M=262144 x N={N}, per env step ONE launch = step + re-deal + masked-uniform "
                                        "sampling + one-hot (M,16,4,4) float32 observation into a torch tensor",
                            "roofline": step_prepare_roofline(262144, ms1, True, True)}
        del sfn, _e
        # (3) the fused rollout at C4's per-GPU shard size (M=262144: the large-M pass of the kernel)
        args.event_every = 4
        sfn, stm, _e = make_fused(262144, N, masked, "fused", args.schedule, seed_off=300)
        d1 = measure(sfn, 120, 20, finish=_e.join_ids)
        ms1 = timers_avg_ms(stm)
        extras["fused_262144"] = {"value": 262144 * N * 120 / d1, "unit": "board-steps/s", "ms_per_step": d1 / 120 * 1e3,
                                  "workload": f"M=262144 (BASELINE config 4's per-GPU shard) x N={N}, fused rollout",
                                  "roofline": rollout_roofline(262144, N, ms1, masked, False)}
        del sfn, _e
        args.event_every = saved_every
        # (4) the fused C2 workload in rocRAND spawn mode (no host PCG64 schedule; statistically the same game,
        # bit-exact with the oracle's counter-based spawn, NOT with numba)
        args.event_every = 8
        sfn, stm, _e = make_fused(M, N, masked, "fused", args.schedule, seed_off=400, spawn="rocrand")
        d1 = measure(sfn, 200, 20, finish=_e.join_ids)
        ms1 = timers_avg_ms(stm)
        extras["rocrand_spawn_mode"] = {"value": M * N * 200 / d1, "unit": "board-steps/s", "ms_per_step": d1 / 200 * 1e3,
                                        "roofline": rollout_roofline(M, N, ms1, masked, False, "rocrand"),
                                        "note": "spawn='rocrand': fused rollout with the rocRAND Philox spawn (one word per "
                                                "tile, one Philox block per four steps), no host schedule; not bit-exact "
                                                "with the numba path, the same game statistically (see DESIGN.md)"}
        del sfn, _e
        # (5) the fused C2 workload with the reference RandomPolicy's masking (policy/random.py:17-27: uniform over the
        # VALID actions; the headline draws unmasked uniform actions as SURVEY 8c defines C2) - bit-exact spawn mode
        if not masked:
            sfn, stm, _e = make_fused(M, N, True, "fused", args.schedule, seed_off=500)
            d1 = measure(sfn, 200, 20, finish=_e.join_ids)
            ms1 = timers_avg_ms(stm)
            extras["masked_actions"] = {"value": M * N * 200 / d1, "unit": "board-steps/s", "ms_per_step": d1 / 200 * 1e3,
                                        "workload": f"M={M} x N={N} fused rollout, actions uniform over each board's valid "
                                                    "actions (the reference RandomPolicy's sampling rule), table spawn",
                                        "roofline": rollout_roofline(M, N, ms1, True, False)}
            del sfn, _e
        args.event_every = saved_every

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(M, N, args.seed, args.cpu_seconds)

    per_rank = [rank_info]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_info)
        per_rank = gathered

    if rank == 0:
        tag = "C2: " if (M, N, args.mode, masked) == (65536, 128, "fused", False) else \
              "C3: " if (M, args.mode, masked) == (262144, "obs", True) else ""
        what = {"fused": "one fused rollout launch per bench step",
                "fused-traj": "one fused rollout launch per bench step, every transition written (REPLAY_SPEC)",
                "stepwise": "one ml2048_step_prepare launch per env step (step + re-deal + next RandomPolicy draw)",
                "obs": "one ml2048_step_prepare launch per env step incl. masked sampling + one-hot (M,16,4,4) f32 observation",
                "stepwise3": "reference-shaped per-step API: prepare, sample, step launches per env step"}[args.mode]
        line = {
            "metric": "env board-steps/sec (M boards x N steps), bit-exact vs game_numba semantics",
            "value": value, "unit": "board-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "warmup_steps_run": warmup_steps_run,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{tag}M={M} boards/GPU x N={N} random-action env steps per bench step "
                                   f"({'masked' if masked else 'unmasked'} uniform actions sampled on device, "
                                   f"{'table spawn = bit-exact mode' if args.spawn == 'table' else 'rocRAND spawn mode'}), {what}",
                       "boards_per_gpu": M, "env_steps_per_step": N, "mode": args.mode, "sharding": f"replicas x{world}",
                       "host_schedule": args.schedule, "ML2048_HOST_THREADS": int(os.environ["ML2048_HOST_THREADS"]),
                       "per_rank": per_rank, "id_scan": args.ids,
                       "spawn": "table (bit-exact with the reference's PCG64 tables)" if args.spawn == "table" else
                                "rocrand (rocRAND Philox spawn: the same game statistically, not bit-exact with numba)",
                       "timing": "W warm-up steps, repeated until --min-warmup-ms (default 30 ms) have passed - the GPU's clocks "
                                 "need ~3 ms of work to ramp up from idle, 40 bench steps of this workload; warmup_steps_run "
                                 "is the count - (the first one carries a kernel timer, so the runtime's one-off set-up for "
                                 "timed dispatches is not inside the timed region), barrier + synchronize, exactly K steps "
                                 "+ the last rollout's deferred game-id scan (VecGame.join_ids), barrier + synchronize; gc.collect() + gc.freeze() before the warm-up; the kernel timer "
                                 "rides on every max(8, K/32)-th timed launch"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extras)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
