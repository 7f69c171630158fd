#!/bin/bash
# Round 4: kept node states in the lane searches (option lane_pool: -1 = by class in the heads and the three-call class, 0 = none, k = k slots everywhere)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lane.py -x -q -m gpu 2>&1 | tail -3
bash tools/r04_sweep1.sh - lane_pool=0 lane_pool=1 lane_pool=2 lane_pool=4 - lane_pool=0 2>&1 | tee gpurun_out/r04_sweep9.txt
bash tools/r03_chain.sh "" 2>&1 | tee gpurun_out/r04_chain_pool.txt
bash tools/r03_chain.sh "lane_pool=0" 2>&1 | tee gpurun_out/r04_chain_nopool.txt
