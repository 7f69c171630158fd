#!/bin/bash
# Round 4: closed-form restore of zero-cost search nodes in the lanes, with / without the kept node states
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lane.py -x -q -m gpu 2>&1 | tail -3
bash tools/r04_sweep1.sh - lane_pool=0 lane_pool=2 lane_pool=8 - lane_waves_per_cu=16 lane_waves_per_cu=8 lane_head_stream=1 2>&1 | tee gpurun_out/r04_sweep10.txt
bash tools/r03_chain.sh "" 2>&1 | tee gpurun_out/r04_chain_restore.txt
