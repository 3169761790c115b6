#!/bin/bash
# The evidence set of a round on the GPU box (run through gpurun, ~6 GPU-minutes): tools/final_set.sh <tag>
#   gpurun_out/<tag>_bench_line.json             the default bench.py run (CPU baseline + drop-in-caller figures)
#   gpurun_out/prof_<tag>/ + profiles/<tag>_*    tools/profile.sh: kernel stats, timeline, PMC passes
#   gpurun_out/<tag>_gpu_profiler_configs.jsonl  the other BASELINE configurations through gpu_profiler
#   gpurun_out/<tag>_lone_latency.txt            tools/lone_latency.py
set -u
tag=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
# the counter passes first: bench.py replays roofline.traffic from profiles/pmc_traffic.json only while that file
# belongs to the kernel source it runs (kernel_source_sha256), so the file is refreshed before the evidence line is taken
bash tools/profile.sh $tag && cp gpurun_out/prof_$tag/summary/${tag}_pmc.json profiles/pmc_traffic.json && \
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err && echo "bench ok" && \
{ cd $R/metal-msm-gpu-acceleration_amd
  ./gpu_profiler 24 1 gpu_resident 5 --layout ark --warmup 4 --json 2>/dev/null | grep '^{'
  ./gpu_profiler 18 1 gpu_resident 30 --warmup 4 --json 2>/dev/null | grep '^{'
  ./gpu_profiler 18 40 gpu_resident 5 --warmup 2 --json 2>/dev/null | grep '^{'
  ./gpu_profiler 16 40 gpu_resident 5 --warmup 2 --json 2>/dev/null | grep '^{'
  ./gpu_profiler 16 1 cpu 5 --json 2>/dev/null | grep '^{'
} > $R/gpurun_out/${tag}_gpu_profiler_configs.jsonl && echo "configs ok" && \
cd $R && python tools/lone_latency.py 10,12,14,16,18,20,22 30 > gpurun_out/${tag}_lone_latency.txt 2>/dev/null && echo "lone ok" && \
for v in "--persistent-bases" "--precomputed-tables"; do python bench.py $v --no-cpu-baseline --no-extras 2>/dev/null; done > gpurun_out/${tag}_bench_tables_persistent.jsonl && echo "variants ok"
