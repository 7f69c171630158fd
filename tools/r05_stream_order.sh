#!/bin/bash
# round 5: the step against the number of never-used streams made in front of the side streams / behind the wide stream / behind the second lane stream
# (AVK_SPARE_STREAMS=a,b,c: the runtime hands out hardware queues in creation order): whole genome (50 queued steps), rank 0's shard of 8, chr20
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
for sp in "$@"; do
  printf "%-10s genome " "$sp"
  for rep in 1 2; do AVK_SPARE_STREAMS=$sp timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done
  AVK_SPARE_STREAMS=$sp python tools/gpu_shard_step.py 8 - 2>&1 | tail -1 | sed 's/^- *//; s/(lanes[^)]*)//g'
done
