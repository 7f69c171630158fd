#!/bin/bash
# Round 4: a longer parity sweep on the final build with seeds no earlier sweep used
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
BUDGET_S=${1:-600} SEED_BASE=500000 timeout 1500 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_long.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_gpu_fuzz_long.txt
BUDGET_S=${2:-420} SEED_BASE=510000 AVK_OPTS=class_c_nodes_x2=1000 timeout 1200 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_long_classc.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_gpu_fuzz_long_classc.txt | cut -c1-200
BUDGET_S=${3:-300} SEED_BASE=520000 AVK_OPTS=lane_pool=1,wide_lds_bytes=8192,lane_head_width=4 timeout 900 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_long_opts.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_gpu_fuzz_long_opts.txt | cut -c1-200
