"""Timeline analysis of a rocprofv3 --kernel-trace CSV: per kernel duration, and the idle gap in front of each
k_rollout launch (end of the previous kernel on the device -> start of k_rollout).

    python tools/trace_gaps.py <dir containing *_kernel_trace.csv>
"""
import csv
import os
import re
import glob
import statistics as st
import sys

f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = {}
for s, e, n in rows:
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_[A-Za-z]+|[a-z_]+_kernel)", n)
    short = m.group(1) if m else n[:40]
    names.setdefault(short, []).append(e - s)
for n, d in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:28s} n={len(d):6d} avg {st.mean(d)/1e3:9.2f} us  med {st.median(d)/1e3:9.2f} us  total {sum(d)/1e6:9.3f} ms")
idx = [i for i, r in enumerate(rows) if "k_rollout" in r[2]]
gaps, periods, tails = [], [], []
for a, b in zip(idx[:-1], idx[1:]):
    periods.append(rows[b][0] - rows[a][0])
    last_end = max(r[1] for r in rows[a:b])
    gaps.append(rows[b][0] - last_end)
    tails.append(last_end - rows[a][1])
if periods:
    h = len(periods) // 2
    print("k_rollout start-to-start: median %.1f us (second half: %.1f us)" % (st.median(periods) / 1e3, st.median(periods[h:]) / 1e3))
    print("  work after k_rollout until the stream's last kernel ends: median %.1f us" % (st.median(tails) / 1e3))
    print("  idle before the next k_rollout: median %.1f us" % (st.median(gaps) / 1e3))
