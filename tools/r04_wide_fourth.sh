#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -5 | tee gpurun_out/r04_wide_tests.txt
AVK_LIB=libaardvark_amd_widetiming.so python tools/gpu_wide_phases.py 1.0 2>&1 | tee gpurun_out/r04_wide_phases4.txt
g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
bash tools/r03_chain.sh "-" 2>&1 | tee gpurun_out/r04_chain_fourth.txt
bash tools/r04_sweep1.sh - wide_kernel=0 wide_lane_handbacks=0 tail_priority=0 wide_lazy_blocks=64 wide_lazy_blocks=256 - wide_kernel=0 2>&1 | tee gpurun_out/r04_sweep5.txt
