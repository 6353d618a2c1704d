"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> [out.json key-suffix]

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
Narrower accesses are uncalibrated - both raw and corrected figures are kept.
"""
import collections
import csv
import glob
import json
import re
import statistics as st
import sys


def load(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m:
            continue
        out[m.group(1)].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                int(r["Grid_Size"])))
    return out


def main():
    fe, wr = load(sys.argv[1]), load(sys.argv[2])
    res = {}
    for k in sorted(fe):
        fv = [v for v, _, _ in fe[k]]
        wv = [v for v, _, _ in wr.get(k, [(0.0, 0, 0)])]
        dur = [d for _, d, _ in fe[k]]
        fetch_kib, write_kib = st.median(fv), st.median(wv)
        res[k] = {
            "launches": len(fv), "grid": fe[k][0][2], "median_duration_us_under_pmc": st.median(dur) / 1e3,
            "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
            "hbm_bytes_per_launch_corrected": int((2 * fetch_kib + write_kib) * 1024),
            "hbm_bytes_per_launch_raw": int((fetch_kib + write_kib) * 1024),
        }
        print(f"{k:22s} n={len(fv):4d} grid={fe[k][0][2]:9d} FETCH={fetch_kib:12.1f} KiB  WRITE={write_kib:12.1f} KiB  "
              f"corrected={(2*fetch_kib+write_kib)/1024:10.2f} MiB  dur={st.median(dur)/1e3:8.1f} us")
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)


main()
