#!/bin/bash
# third sweep of round 3 (pair lookup, row-wise lane aligner, het-count prediction, 512-workgroup class C launch in place): 50 queued whole-genome resident steps per setting
cd "$(dirname "$0")/.."
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-52s " "$1"; timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1 ms for 50 steps/'; }
run "lane_kernel=1"
run "lane_kernel=1"
for v in 16 24 48 64; do run "lane_node_cap=$v"; done
for v in 9 11 13; do run "lane_max_est=$v"; done
for v in 8 32; do run "lane_head_width=$v"; done
for v in 8 32; do run "lane_width_three=$v"; done
for v in 128 512; do run "hbm_early_blocks=$v"; done
for v in 128 256 768; do run "hbm_solo_blocks=$v"; done
for v in 8 16 24; do run "class_c_nodes_x2=$v"; done
for v in 4 6 7; do run "solo_min_variants=$v"; done
for v in 12 20 24; do run "waves_per_cu=$v"; done
for v in 8 10 16; do run "lane_waves_per_cu=$v"; done
for v in 1 2 8; do run "pair_blocks_per_cu=$v"; done
run "lane_head_stream=1"
run "lane_max_calls=2"
run "lane_node_cap=48,hbm_early_blocks=512"
run "lane_kernel=1"
