#!/bin/bash
# Round 4: the launch chain of one step of rank 0's shard of an 8-rank job (446 k regions), defaults and with every lane class on
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
[ -f /tmp/wshard.bin ] || python tools/dump_workload.py 1.0 /tmp/wshard.bin 50 0 8 > /dev/null
W=/tmp/wshard.bin bash tools/chain_timeline.sh "-" "lane_min_regions=512" 2>&1 | tee gpurun_out/r04_shard_chain.txt
