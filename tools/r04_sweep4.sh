#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
bash tools/r04_sweep1.sh - wide_kernel=0 lane_head_stream=1 lane_waves_per_cu=16 lane_waves_per_cu=8 lane_waves_three=6 lane_waves_three=4 lane_node_cap=48 lane_node_cap=64 wide_blocks=384 wide_lds_bytes=12288 wide_lds_bytes=24576 pair_blocks_per_cu=2 pair_blocks_per_cu=8 waves_per_cu=12 lane_width_three=8 lane_width_three=32 lane_head_width=8 lane_head_width=32 hbm_early_blocks=64 tail_priority=0 lane_head_stream=1,lane_node_cap=48 2>&1 | tee gpurun_out/r04_sweep4.txt
