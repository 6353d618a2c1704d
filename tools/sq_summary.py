"""Summarise rocprofv3 --pmc SQ_* passes for the fused rollout kernel: counter totals per launch and per
wave-step (one wave advancing its 64 boards by one env step).

    python tools/sq_summary.py [--M boards] [--N steps] <pass dir> [<pass dir> ...]   (default M = 65536, N = 128)
"""
import collections
import csv
import glob
import json
import statistics as st
import sys

M, N = 65536, 128
argv = sys.argv[1:]
while argv and argv[0] in ("--M", "--N"):
    if argv[0] == "--M":
        M = int(argv[1])
    else:
        N = int(argv[1])
    argv = argv[2:]
sys.argv[1:] = argv
waves = (M + 63) // 64
acc = collections.defaultdict(list)
dur = []
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = set()
out = {"kernel": "k_rollout", "M": M, "N": N, "waves": waves,
       "launch_us_under_pmc": st.median(dur) / 1e3 if dur else None, "counters": {}}
for name, vals in sorted(acc.items()):
    v = st.median(vals)
    out["counters"][name] = {"per_launch": v, "per_wave_step": v / (waves * N), "launches": len(vals)}
out["note"] = ("SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycles (4 clocks) summed over waves; "
               "per_wave_step divides by waves x N")
print(json.dumps(out, indent=1))
