#!/bin/bash
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_merge
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT" -o merge -- python3 $ROOT/tools/gpu_merge_timing.py > "$OUT/run.log" 2>&1
echo "rc $?"; tail -4 "$OUT/run.log" | cut -c1-300
ls -la "$OUT"
