#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for share in 0.001; do
  echo "=== largest $share of the --min-variant-gap 1000 regions"
  GAP=1000 LARGEST=$share AVK_LIB=libaardvark_amd_phasetiming.so timeout 600 python tools/gpu_wave_phases.py 0.05 2>&1 | tail -17
done | tee gpurun_out/r04_gap_phases.txt
