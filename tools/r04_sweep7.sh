#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
B="tail_priority=0,wide_lazy_blocks=512"
bash tools/r04_sweep1.sh $B $B,lane_head_stream=1 $B,wide_blocks=768 $B,wide_blocks=384 tail_priority=0,wide_lazy_blocks=1024 $B,lane_node_cap=24 $B,lane_node_cap=16 $B,het_search_min=5 $B,lane_max_est=8 $B,lane_max_est=4 $B,lane_head_est=2 $B,hbm_solo_blocks=64 $B wide_kernel=0 2>&1 | tee gpurun_out/r04_sweep7.txt
