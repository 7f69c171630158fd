#!/bin/bash
# the quarter-genome clustered with --min-variant-gap 1000: queued resident steps, polled; and the resident step with the two packers
cd "$(dirname "$0")/.."
python tools/dump_workload.py 0.25 /tmp/wgap.bin 1000
python tools/dump_workload.py 1.0 /tmp/w100.bin
for steps in 1 8; do
  echo "== gap1000 steps $steps"; AVK_TIMING=1 timeout 200 .scratch/first_step_probe /tmp/wgap.bin $steps 0 60 2>&1 | tail -4
done
echo "== gap1000, fixed 1 MB slices, 1 step"; timeout 200 .scratch/first_step_probe /tmp/wgap.bin 1 0 30 adaptive_ws=0 2>&1 | tail -2
for dp in 1 0 1 0; do
  echo "== genome, device_pack=$dp, 50 steps"; timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 0 25 device_pack=$dp 2>&1 | tail -1
done
