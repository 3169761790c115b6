"""Gaps between consecutive accumulate launches in a rocprofv3 --kernel-trace CSV (is the main stream kept busy?).
usage: acc_gaps.py <kernel_trace.csv>"""
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "accumulate_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
print(f"{len(rows)} accumulate launches: duration median {statistics.median(durs):.1f} us")
g = sorted(gaps)
print(f"gaps between them: median {statistics.median(gaps):.1f} us, p25 {g[len(g) // 4]:.1f}, p75 {g[3 * len(g) // 4]:.1f}, max {g[-1]:.1f}")
print("last 12 gaps:", " ".join(f"{x:.0f}" for x in gaps[-12:]))
