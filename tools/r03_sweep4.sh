#!/bin/bash
# LDS shares of the lane launches (round 3, final build): 50 queued whole-genome resident steps per setting, three runs each
cd "$(dirname "$0")/.."
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-52s " "$1"; for k in 1 2 3; do timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo; }
run "lane_kernel=1"
for v in 4 5 6 7; do run "lane_waves_three=$v"; done
for v in 8 10; do run "lane_waves_per_cu=$v,lane_waves_three=5"; done
run "lane_waves_per_cu=8"
run "hbm_solo_blocks=128"
run "hbm_solo_blocks=128,lane_waves_three=5"
run "lane_kernel=1"
