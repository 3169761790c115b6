#!/bin/bash
# Shader clock and power while the headline bench runs (rocm-smi sampled every 0.5 s): tools/sample_clocks.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --no-cpu-baseline --no-extras --steps 400 > gpurun_out/clk_bench.json 2>/dev/null &
pid=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo
  sleep 0.5
done
wait $pid
python -c "
import json
d=json.loads(open('gpurun_out/clk_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['roofline']['avg_launch_ms'])"
echo "idle:"; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
