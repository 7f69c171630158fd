#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
bash tools/r04_sweep1.sh - wide_retry_lds_bytes=0 lane_wide_est=12 lane_wide_est=8 lane_wide_est=6 lane_wide_est=4 lane_wide_est=3 lane_wide_est=2 lane_wide_est=4,wide_blocks=1024 lane_wide_est=2,wide_blocks=1024 lane_wide_est=1,wide_blocks=1536 - wide_kernel=0 2>&1 | tee gpurun_out/r04_sweep8.txt
bash tools/r03_chain.sh "lane_wide_est=4,wide_blocks=1024" 2>&1 | tee gpurun_out/r04_chain_sixth.txt
