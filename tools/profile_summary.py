"""Condenses the rocprofv3 outputs of tools/profile.sh into the small files kept under profiles/:
   <tag>_bench_kernel_stats.csv   the --stats kernel table
   <tag>_timeline.txt             two consecutive instances in steady state, from the kernel trace
   <tag>_pmc.json                 per-kernel FETCH_SIZE / WRITE_SIZE (KB per dispatch) and SQ counters (per dispatch)"""
import csv
import glob
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"msm_amd::", "", name)
    return re.sub(r"\(.*$", "", name)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return hits[0] if hits else None


# ---- kernel stats table
f = find("stats/**/*kernel_stats.csv")
if f:
    with open(f) as fh, open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w") as oh:
        oh.write(fh.read())

# ---- timeline of two steady-state instances
f = find("stats/**/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows))
    acc = [i for i, e in enumerate(ev) if e[2].startswith("accumulate_kernel")]
    if len(acc) > 40:
        a0 = acc[len(acc) // 2]
        t0 = ev[a0][0]
        a2 = acc[len(acc) // 2 + 2]
        lo = t0 - 700_000
        hi = ev[a2][0]
        with open(os.path.join(dst, f"{tag}_timeline.txt"), "w") as oh:
            oh.write(f"# rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline --no-extras` ({tag})\n")
            oh.write("# two consecutive instances in steady state; microseconds from the start of an accumulate\n")
            oh.write("#   start       end       dur  kernel\n")
            for s, e, n in ev:
                if lo <= s < hi:
                    oh.write(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}\n")
        gaps = [ev[acc[i + 1]][0] - ev[acc[i]][1] for i in range(len(acc) // 4, len(acc) - 1)]
        durs = [ev[i][1] - ev[i][0] for i in acc[len(acc) // 4:]]
        with open(os.path.join(dst, f"{tag}_timeline.txt"), "a") as oh:
            oh.write(f"# accumulate launches: mean duration {sum(durs) / len(durs) / 1e3:.1f} us, mean gap to the next "
                     f"{sum(gaps) / len(gaps) / 1e3:.1f} us, period {(sum(durs) / len(durs) + sum(gaps) / len(gaps)) / 1e3:.1f} us\n")

# ---- counters
pmc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = find(os.path.relpath(d, out) + "/**/*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        v = float(r["Counter_Value"])
        s = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
        s[0] += v
        s[1] += 1
    for k, cs in acc.items():
        for c, (tot, cnt) in cs.items():
            pmc.setdefault(k, {})[c] = round(tot / cnt, 1)
            pmc[k]["dispatches_" + c] = cnt
if pmc:
    a = next((v for k, v in pmc.items() if k.startswith("accumulate_kernel")), None)
    summary = {"source": "tools/profile.sh: rocprofv3 --kernel-trace --pmc <counter> (one pass per counter group), "
                         "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras",
               "unit": "FETCH_SIZE / WRITE_SIZE in KB per dispatch, SQ_* raw per dispatch (averaged over dispatches)",
               "calibration": "FETCH_SIZE tallies a 128-byte request as 64 B on gfx950 (MI355X_MICROARCH.md, HBM section; "
                              "tools/microbench/gather_calib.hip): bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact",
               "per_kernel": pmc}
    if a and "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        summary["accumulate_log20_bytes_per_launch"] = int((2 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024)
    # which kernel source these counters belong to: bench.py refuses roofline.traffic when the files have changed since
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    summary["kernel_source_sha256"] = bench.kernel_source_digest()
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
print("summary files:", sorted(os.listdir(dst)))
