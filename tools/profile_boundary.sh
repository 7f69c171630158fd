#!/bin/bash
# kernel + memory-copy trace of avk_compare_batch on the benchmark genome (pinned arrays): tools/profile_boundary.sh <tag>
TAG=${1:-r03_boundary}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT" -o "$TAG" -- python3 $ROOT/tools/boundary_once.py 1.0 3 > "$OUT/run.log" 2>&1
echo "trace rc $?"; tail -5 "$OUT/run.log"
python3 $ROOT/tools/summarize_boundary.py "$OUT/${TAG}_results.db" "$ROOT/gpurun_out/${TAG}_timeline.txt" | tail -70
