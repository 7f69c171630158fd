#!/bin/bash
# On the GPU box: for each option string (one argument each; "-" = defaults): wall time of 50 queued whole-genome resident steps in a fresh
# process, then the kernel timeline of one step (rocprofv3 --kernel-trace on the same probe, 3 steps): which launch ends when.
# usage: tools/chain_timeline.sh "-" "lane_quad=0" ...
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
W=${W:-/tmp/w100.bin}   # W=/tmp/wshard.bin (python tools/dump_workload.py 1.0 /tmp/wshard.bin 50 0 8): rank 0's shard of an 8-rank job
[ -f $W ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
k=0
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  k=$((k+1))
  echo "=== options: ${o:-defaults}"
  for rep in 1 2; do timeout 120 .scratch/first_step_probe $W 50 1 25 "$o" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/50 steps: \1 ms/'; done
  ( cd /tmp && export TMPDIR=/tmp AVK_PROBE_TEARDOWN=1 && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/chain_$k -o chain -- $R/.scratch/first_step_probe $W 3 1 25 "$o" > $R/gpurun_out/chain_$k.log 2>&1 )
  db=$(ls gpurun_out/chain_$k/*/chain_results.db gpurun_out/chain_$k/chain_results.db 2>/dev/null | head -1)
  python tools/show_timeline.py $db 2>&1 | tail -22
  rm -rf gpurun_out/chain_$k
done
