#!/bin/bash
# round 5: phase shares of the quad launches (timing build), then 50-step times per option string (arguments; "-" = defaults)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TAG=${TAG:-r05}
AVK_LIB=libaardvark_amd_lanetiming.so timeout 600 python tools/gpu_lane_phases.py 1.0 > gpurun_out/${TAG}_phases.txt 2>&1
cat gpurun_out/${TAG}_phases.txt
tools/sweep_options.sh "$@" > gpurun_out/${TAG}_sweep.txt 2>&1
cat gpurun_out/${TAG}_sweep.txt
