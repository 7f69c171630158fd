#!/bin/bash
# Round 4: where the two launches that end a step spend their lanes' time (profiling build of the lane kernel)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
AVK_LIB=libaardvark_amd_lanetiming.so timeout 600 python tools/gpu_lane_phases.py 1.0 2>&1 | tail -24 | tee gpurun_out/r04_lane_phases.txt
