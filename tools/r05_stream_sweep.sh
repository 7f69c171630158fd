#!/bin/bash
# round 5: 50 queued whole-genome steps for every placement of up to 3 never-used streams behind the wide stream / the second lane stream / the third lane stream
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
W=${W:-/tmp/w100.bin}
for a in 0 1 2 3; do for b in 0 1 2 3; do for c in 0 1 2; do
  sp="0,$a,$b,$c"
  printf "%-10s " "$sp"
  for rep in 1 2; do AVK_SPARE_STREAMS=$sp timeout 120 .scratch/first_step_probe $W 50 1 25 "" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo
done; done; done
