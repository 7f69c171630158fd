#!/bin/bash
# Collects the rocprofv3 evidence for one round: kernel-trace stats of the default bench command and
# PMC passes (each in its own run).  Usage (on the GPU box, from the repo root): tools/profile_gpu.sh r01
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o "$TAG" -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats rc $?"
SHORT="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o "$TAG" -- $SHORT > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i ($set) rc $?"
done
find "$OUT" -name "*.csv" | head -40
