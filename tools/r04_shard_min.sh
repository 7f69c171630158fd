#!/bin/bash
# Round 4: rank 0's shard of 8 under lower launch thresholds of the lane classes, on the device's clock (AVK_TIMING: chains of the launch graph) and as wall time per boundary call
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
for rep in 1 2; do for o in "" lane_min_regions=1024 lane_min_regions=512 lane_min_regions=256; do
  echo "== options: ${o:-defaults}"
  timeout 300 python3 tools/r04_shard_host.py 8 "$o" 2>&1 | grep -E "chains end|^call|class B" | tail -9 | sed 's/avk compare packed, //; s/avk upload (device-packed): //' | cut -c1-260
done; done | tee gpurun_out/r04_shard_min.txt
