#!/bin/bash
# round 5: placements of the never-used streams against each other on ONE box, interleaved: 50 queued whole-genome steps each, N rounds
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
N=${N:-5}
for round in $(seq 1 $N); do
  for sp in "$@"; do
    t=$(AVK_SPARE_STREAMS=$sp timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/')
    echo "$sp $t"
  done
done | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
