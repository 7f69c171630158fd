#!/bin/bash
# Round 4: the lane kernel compiled for 2 / 3 (default) / 4 waves per SIMD (168 / 256 / 128 VGPRs): resident step and boundary call on one box
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for lib in "" libaardvark_amd_wpe2.so libaardvark_amd_wpe4.so ""; do
  echo "=== ${lib:-default (3 waves per SIMD)}"
  env ${lib:+AVK_LIB=$lib} timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-parity --steps 40 --resident-steps 200 2>&1 | grep -E "timed region|resident leg"
done | tee gpurun_out/r04_wpe.txt
