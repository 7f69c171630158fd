#!/bin/bash
# kernel + memory-copy trace of whole genomes back to back through the asynchronous boundary: tools/profile_pipeline.sh <tag> [options]
TAG=${1:-r05_boundary}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT" -o "$TAG" -- python3 $ROOT/tools/boundary_pipelined.py 1.0 8 "$2" ${3:-2} > "$OUT/run.log" 2>&1
echo "trace rc $?"; tail -2 "$OUT/run.log"
python3 $ROOT/tools/summarize_pipeline.py $(ls $OUT/*/${TAG}_results.db $OUT/${TAG}_results.db 2>/dev/null | head -1) "$ROOT/gpurun_out/${TAG}_timeline.txt" | tail -70
