#!/bin/bash
# the class-C-everywhere sweep's first cases under options, twice each: which switch makes the fourth case (seed 411003) slow and wrong?
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for o in "wide_kernel=0" "wide_retry_lds_bytes=0" "adaptive_ws=0" "lane_pool=0" "wide_lane_handbacks=0" ""; do
  for rep in 1 2; do
    echo "=== class_c_nodes_x2=1000,$o"
    BUDGET_S=12 FUZZ_N=20000 SEED_BASE=410000 AVK_OPTS=class_c_nodes_x2=1000,$o timeout 100 python tools/gpu_fuzz.py 2>&1 | grep -E "fuzz seed 41100[0-9]|bad regions|ALL OK" | cut -c1-75,95-260
  done
done 2>&1 | tee gpurun_out/r04_hang.txt
