#!/bin/bash
# resident whole-genome step under one scheduling option at a time (50 steps queued back to back, fresh process each): tools/r03_sweep.sh > gpurun_out/r03_sweep.txt
cd "$(dirname "$0")/.."
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-44s " "$1"; timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 0 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1 ms for 50 steps/'; }
run "lane_kernel=1"
run "lane_kernel=1"
for v in 3 5 7 9 11; do run "lane_max_est=$v"; done
for v in 16 24 48 64; do run "lane_node_cap=$v"; done
for v in 8 32; do run "lane_head_width=$v"; done
for v in 8 32; do run "lane_width_three=$v"; done
for v in 128 512; do run "hbm_early_blocks=$v"; done
for v in 64 256; do run "hbm_solo_blocks=$v"; done
for v in 4 6 7; do run "solo_min_variants=$v"; done
for v in 8 16 24; do run "class_c_nodes_x2=$v"; done
for v in 8192 12288 16384; do run "lds_bytes_per_wave=$v"; done
for v in 12 20 24; do run "waves_per_cu=$v"; done
for v in 8 16 20; do run "lane_waves_per_cu=$v"; done
for v in 2; do run "lane_max_calls=$v"; done
run "lane_head_stream=1"
run "lane_max_est=7,lane_node_cap=24"
run "lane_kernel=1"
