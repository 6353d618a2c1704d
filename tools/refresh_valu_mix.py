"""Regenerate profiles/valu_mix.json: compile the three plain passes of ml2048_hip.hip to ISA (as ml2048_amd/build.py
does), run tools/valu_mix.py on the kernel k_rollout<normal, table, plain, INV> of each, and map its step loops onto the
entries bench.py prices (a loop is recognised by its look-ups and its Philox share: see `pick`).

    python tools/refresh_valu_mix.py            # needs hipcc only (no GPU)
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "ml2048_amd", "csrc", "ml2048_hip.hip")
RATE = os.path.join(ROOT, "profiles", "r03_valu_rate.log")
OUT = os.path.join(ROOT, "profiles", "valu_mix.json")
SYM = "k_rolloutILi0ELb0ELb0ELb1"
PASSES = {  # tag -> extra flags (build.py)
    "ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DML2048_LONE_WAVE=1"],
    "lone": ["-DML2048_LONE_WAVE=1"],
    "dflt": [],
}


def loops_of(tag, tmp):
    asm, js = os.path.join(tmp, tag + ".s"), os.path.join(tmp, tag + ".json")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           f"-DML2048_ROLLOUT_PLAIN_TU=ml2048_internal_launch_rollout_plain_{tag}", *PASSES[tag],
                           "--cuda-device-only", "-S", "-o", asm, SRC], stderr=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "valu_mix.py"), asm, SYM, RATE, js],
                          stdout=subprocess.DEVNULL)
    return list(json.load(open(js))["loops"].values())


def pick(loops, table, masked):
    """the loop with / without line-table look-ups; the masked loops carry a Philox block per step in their body"""
    c = [l for l in loops if (l["lds_reads_per_step"] > 0) == table and (l["philox_mads_per_step"] > 10) == masked]
    assert c, (table, masked, loops)
    return min(c, key=lambda l: l["valu_static_per_step"])  # (the reward-carrying masked loop is the longer one)


with tempfile.TemporaryDirectory() as tmp:
    L = {t: loops_of(t, tmp) for t in PASSES}
old = json.load(open(OUT))
new = {"_doc": old["_doc"]}
plan = [("k_rollout:plain:small", "lone", True, False), ("k_rollout:masked:small", "ilp", True, True),
        ("k_rollout:plain:large", "dflt", True, False), ("k_rollout:masked:large", "dflt", True, True),
        ("k_rollout:plain:large:no-table", "dflt", False, False), ("k_rollout:plain:small:no-table", "lone", False, False)]
for key, tag, table, masked in plan:
    e = dict(pick(L[tag], table, masked))
    e.pop("top", None)
    e["loop"] = old[key]["loop"]
    new[key] = e
    print(key, e["valu_static_per_step"], e["half_rate_share"], e["mean_cost_vs_full_rate"],
          "(was %s %s)" % (old[key]["valu_static_per_step"], old[key]["mean_cost_vs_full_rate"]))
json.dump(new, open(OUT, "w"), indent=1)
