"""Timeline of the LAST lone call in a rocprofv3 --kernel-trace CSV (development aid).
usage: lone_timeline.py <kernel_trace.csv> [n_kernels_back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: walk back from the last kernel to the previous convert_bases_kernel
last = len(rows) - 1
first = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("convert_bases") or "convert_bases" in r["Kernel_Name"])
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
busy = 0
print("#   start      end      dur   gap_before  kernel")
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("msm_amd::", "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:10.1f}  {name}")
    busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
print(f"# span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
