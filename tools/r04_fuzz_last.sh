#!/bin/bash
# Round 4: one more parity sweep on the final build: default options, every eligible region planned as class C, and this round's two kernel changes switched off (seeds 600000 / 610000 / 620000)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
BUDGET_S=700 SEED_BASE=600000 timeout 1500 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_last.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_gpu_fuzz_last.txt | cut -c1-300
BUDGET_S=400 SEED_BASE=610000 AVK_OPTS=class_c_nodes_x2=1000 timeout 1200 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_last_classc.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_gpu_fuzz_last_classc.txt | cut -c1-300
BUDGET_S=300 SEED_BASE=620000 AVK_OPTS=lane_pool=0,wide_kernel=0 timeout 900 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_last_off.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_gpu_fuzz_last_off.txt | cut -c1-300
