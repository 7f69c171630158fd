#!/bin/bash
# usage (GPU box): tools/timeline_run.sh <tag> <python script and args...>  -> prints the launch timeline of the last steps
TAG=$1; shift
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/$TAG -o tl -- python3 $R/"$@" > /dev/null 2>&1
cd $R; python3 tools/timeline_stats.py $(find gpurun_out/$TAG -name "*.db" | head -1)
