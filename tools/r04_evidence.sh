#!/bin/bash
# Round 4: everything under profiles/r04_* from one box, on the final build: kernel stats + PMC passes of the bench command, the boundary call's timeline,
# the A/B of the round's kernel changes, one step's launch chain, lane phases, the large-window leg's tail, then the default bench run
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
bash tools/r04_profile.sh r04 2>&1 | tail -14
cd $R
AVK_TIMING=1 timeout 300 python3 tools/boundary_once.py 1.0 4 2>&1 | grep -v "avk pool" | tail -10 > gpurun_out/r04_boundary_host.txt
bash tools/profile_boundary.sh r04_boundary > /dev/null 2>&1; find gpurun_out/prof_r04_boundary -name "*.db" -delete
cd $R
bash tools/sweep_options.sh - lane_pool=0 wide_kernel=0 lane_pool=0,wide_kernel=0 - 2>&1 | tee gpurun_out/r04_sweep_final.txt
bash tools/chain_timeline.sh "-" 2>&1 | tee gpurun_out/r04_chain.txt
bash tools/r04_lane_phases.sh > /dev/null 2>&1
timeout 900 python tools/r04_gap_tail.py 2>&1 | tail -8 > gpurun_out/r04_gap_tail.txt
timeout 1800 python bench.py 2> gpurun_out/r04_bench.log > gpurun_out/r04_bench.json; echo "bench rc $?"; tail -14 gpurun_out/r04_bench.log
