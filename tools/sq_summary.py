"""Summarise rocprofv3 --pmc SQ_* passes for the fused rollout kernel: counter totals per launch and per
wave-step (one wave advancing its 64 boards by one env step).

    python tools/sq_summary.py <pass dir> [<pass dir> ...]      (M = 65536, N = 128 assumed: bench.py default)
"""
import collections
import csv
import glob
import json
import statistics as st
import sys

M, N = 65536, 128
waves = M // 64
acc = collections.defaultdict(list)
dur = []
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"kernel": "k_rollout<0,false,false>", "M": M, "N": N, "waves": waves,
       "launch_us_under_pmc": st.median(dur) / 1e3 if dur else None, "counters": {}}
for name, vals in sorted(acc.items()):
    v = st.median(vals)
    out["counters"][name] = {"per_launch": v, "per_wave_step": v / (waves * N), "launches": len(vals)}
out["note"] = ("SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycles (4 clocks) summed over waves; "
               "per_wave_step divides by waves x N")
print(json.dumps(out, indent=1))
