#!/bin/bash
# Round 4: the parity sweep through the C-ABI on the final build (wide arrays, packed result form, compact and packed batch forms), then again with every region outside
# the lane classes planned as class C (the wide kernel sees it first; adaptive workspaces at their largest).  usage: tools/r04_fuzz.sh [seconds] [seconds]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
if [ "${1:-330}" != "0" ]; then
BUDGET_S=${1:-330} SEED_BASE=400000 timeout 900 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_report.txt 2>&1; echo "rc $?"; tail -2 gpurun_out/r04_gpu_fuzz_report.txt
fi
BUDGET_S=${2:-240} SEED_BASE=410000 AVK_OPTS=class_c_nodes_x2=1000 timeout 700 python tools/gpu_fuzz.py > gpurun_out/r04_gpu_fuzz_classc.txt 2>&1; echo "rc $?"; tail -2 gpurun_out/r04_gpu_fuzz_classc.txt | cut -c1-300
