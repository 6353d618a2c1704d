"""Copy the judged summaries of one collected profile set (tools/collect_profiles.sh <tag> ..., merged back into
gpurun_out/<tag>/) into profiles/ under the tag's name, and refresh the tables bench.py prices its rooflines with:
profiles/valu_insts.json (SQ counters of the rollout kernel's variants) and profiles/traffic.json (PMC HBM bytes).

    python tools/install_profiles.py r02_s            # whatever parts gpurun_out/r02_s holds
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
done = []


def put(name_in, name_out):
    a = os.path.join(src, name_in)
    if os.path.exists(a):
        shutil.copyfile(a, os.path.join(dst, name_out))
        done.append(name_out)
        return True
    return False


for n in ("bench_default.json", "bench_driver_style.json", "bench_default_under_rocprof.json", "gputests.log", "soak_modes.log"):
    put(n, f"{tag}_{n}")
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copyfile(f, os.path.join(dst, f"{tag}_bench_default_kernel_stats.csv"))
    done.append(f"{tag}_bench_default_kernel_stats.csv")

# ---- SQ counters -> valu_insts.json
vi_path = os.path.join(dst, "valu_insts.json")
vi = json.load(open(vi_path))
keys = {"plain_small": "k_rollout:plain:small", "plain_large": "k_rollout:plain:large", "masked_small": "k_rollout:masked:small",
        "masked_large": "k_rollout:masked:large", "rocrand_small": "k_rollout:plain-rocrand:small",
        "rocrand_large": "k_rollout:plain-rocrand:large"}
for short, key in keys.items():
    name = f"sq_counters_{short}.json"
    if not put(name, f"{tag}_{name}"):
        continue
    d = json.load(open(os.path.join(src, name)))
    c = {k: v["per_wave_step"] for k, v in d["counters"].items()}
    if "SQ_INSTS_VALU" not in c:
        continue
    all_insts = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"))
    vi[key] = {"valu_per_wave_step": round(c["SQ_INSTS_VALU"], 2), "all_per_wave_step": round(all_insts, 2),
               "wave_quad_cycles_per_step": round(c.get("SQ_WAVE_CYCLES", 0.0), 1),
               "wait_any_quad_cycles_per_step": round(c.get("SQ_WAIT_ANY", 0.0), 1),
               "measured_at": f"M={d['M']}, N={d['N']}", "source": f"profiles/{tag}_{name}"}
json.dump(vi, open(vi_path, "w"), indent=1)

# ---- PMC traffic -> traffic.json
tr_path = os.path.join(dst, "traffic.json")
tr = json.load(open(tr_path))
pm = {"fused_c2": "k_rollout:M=65536:N=128", "fused_262144": "k_rollout:M=262144:N=128",
      "obs_262144": "k_step_prepare:M=262144:obs=1", "stepwise_c2": "k_step_prepare:M=65536:obs=0"}
note = "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2*FETCH+WRITE KiB per MI355X_MICROARCH.md)"
for short, key in pm.items():
    name = f"pmc_{short}_summary.json"
    if not put(name, f"{tag}_pmc_{short}.json"):
        continue
    d = json.load(open(os.path.join(src, name)))
    kern = key.split(":")[0]  # the summary is per kernel name: the one this pass was collected for
    cand = [v for k, v in d.items() if k.startswith(kern) and "hbm_bytes_per_launch_corrected" in v]
    if cand:
        b = max(cand, key=lambda v: v["launches"])["hbm_bytes_per_launch_corrected"]
        tr[key] = {"bytes_per_launch": int(b), "source": f"profiles/{tag}_pmc_{short}.json {note}"}
json.dump(tr, open(tr_path, "w"), indent=1)
print("installed:", *done, sep="\n  ")
