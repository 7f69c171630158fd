#!/bin/bash
# Round 4: the boundary call's device-clock marks (AVK_TIMING) under several in-tree builds of the library, whole genome and rank 0's shard of 8, alternating
# usage: tools/r04_writers_tl.sh <lib> ..
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib"
  AVK_LIB=$lib AVK_TIMING=1 timeout 300 python3 tools/boundary_once.py 1.0 10 2>&1 | grep -E "device clock" | tail -6 | sed 's/avk compare packed, device clock from the first copy: //'
  AVK_LIB=$lib AVK_TIMING=1 timeout 300 python3 tools/r04_shard_host.py 2>&1 | grep -E "device clock" | tail -3 | sed 's/avk compare packed, device clock from the first copy: /shard: /'
done; done | tee gpurun_out/r04_writers_tl.txt
