#!/bin/bash
# Round 4: the hand-back launches behind the last lane chain of a whole-genome call under context options, on the device's clock (AVK_TIMING chain ends)
# usage: tools/r04_tail_sweep.sh <options> ..   ("-" = defaults)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
for rep in 1 2; do for o in "$@"; do
  [ "$o" = "-" ] && o=""
  echo "== options: ${o:-defaults}"
  AVK_TIMING=1 timeout 300 python3 tools/boundary_once.py 1.0 12 "$o" 2>&1 | grep -E "chains end" | tail -8 | sed 's/avk compare packed, chains end (ms after the first solver launch)://' | cut -c1-260
done; done | tee gpurun_out/r04_tail_sweep.txt
