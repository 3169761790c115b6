#!/bin/bash
# BASELINE configurations and drop-in-caller shapes through the C++ CLI (system HIP runtime, no torch in the process)
P=./metal-msm-gpu-acceleration_amd/gpu_profiler
j() { "$@" --json 2>/dev/null | grep '^{'; }
echo "# config 1: gpu_profiler 16 1 cpu 5 (no GPU)"; j $P 16 1 cpu 5 --warmup 1
echo "# config 2: 2^18 x 1, resident"; j $P 18 1 gpu_resident 30 --warmup 4
echo "# config 3: 2^20 x 5, resident (one batched call per pass)"; j $P 20 5 gpu_resident 10 --warmup 2
echo "# config 3 through --gpus 1 (sharded entry point + RCCL gather of one rank)"; j $P 20 10 gpu_resident 3 --warmup 1 --gpus 1 --devices 0
echo "# the same ten instances, single context"; j $P 20 10 gpu_resident 3 --warmup 1
echo "# two contexts sharing the GPU"; j $P 20 10 gpu_resident 3 --warmup 1 --devices 0,0
echo "# drop-in caller: 5 x 2^20 host slices per pass, pageable (parallel=true -> msm_amd_msm_batch)"; j $P 20 5 gpu 5 true --warmup 1
echo "# the same with the bases cache"; j $P 20 5 gpu 5 true --warmup 1 --bases-cache 1024
echo "# sequential blocking calls (gpu_msm_h2c), pageable"; j $P 20 5 gpu 5 --warmup 1
echo "# the same with the bases cache"; j $P 20 5 gpu 5 --warmup 1 --bases-cache 1024
echo "# msm_best per instance"; j $P 20 5 best_gpu 5 --warmup 1
echo "# msm_best with the bases cache"; j $P 20 5 best_gpu 5 --warmup 1 --bases-cache 1024
echo "# gpu_cpu with the measured split and with the reference's"; $P 20 1 gpu_cpu 3 --warmup 1 2>&1 | grep -E "split|Average"; $P 20 1 gpu_cpu 3 --warmup 1 --reference-split 2>&1 | grep -E "split|Average"
echo "# check mode (gpu_with_cpu at the reference's split == cpu)"; $P 18 2 check 1 2>&1 | grep -E "ERROR|Average"
