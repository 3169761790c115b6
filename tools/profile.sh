#!/bin/bash
# Profiling passes of the headline bench on the GPU box (run through gpurun; outputs under gpurun_out/prof_<tag>/).
#   tools/profile.sh <tag> [bench args...]
# pass 1: rocprofv3 --kernel-trace --stats over the default bench run (kernel table + trace for the timeline)
# pass 2..4: PMC counters in their own short runs (FETCH_SIZE / WRITE_SIZE cannot share a pass; SQ counters)
# The program after `--` is python3 itself (no env/bash hop: the profiler has initialised the GPU already).
set -u
tag=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $out/bench_line_under_rocprof.json 2> $out/stats.err
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $out/pmc_$c.err
  echo "pmc $c rc=$?"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc_SQ -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $out/pmc_SQ.err
echo "pmc SQ rc=$?"
python3 $R/tools/profile_summary.py $out $tag
