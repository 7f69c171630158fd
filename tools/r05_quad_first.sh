#!/bin/bash
# round 5, first GPU call of the quad kernel: parity (quads on by default), then the step with and without them
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_quad.py tests/test_gpu_lane.py tests/test_gpu_wide.py -m gpu -x -q > gpurun_out/r05_quad_tests.txt 2>&1
tail -5 gpurun_out/r05_quad_tests.txt
tools/chain_timeline.sh - lane_quad=0 > gpurun_out/r05_quad_chain.txt 2>&1
cat gpurun_out/r05_quad_chain.txt
