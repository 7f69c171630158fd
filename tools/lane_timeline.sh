#!/bin/bash
# kernel timeline of a few bench steps (rocprofv3 --kernel-trace): usage on the GPU box: tools/lane_timeline.sh <tag> [more bench.py flags, e.g. --scale 0.0135 for a chr20-sized batch]
TAG=${1:-tl}
shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG/stats -o $TAG -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --boundary-calls 0 --no-supervisor "$@" > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
echo prof rc $?
grep "bench " $R/gpurun_out/prof_${TAG}_stats.log | cut -c1-400
