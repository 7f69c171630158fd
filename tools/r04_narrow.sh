#!/bin/bash
# Round 4: the three-call class on small batches: launch threshold (x2 of lane_min_regions) and records per wave
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
[ -f /tmp/wshard.bin ] || python tools/dump_workload.py 1.0 /tmp/wshard.bin 50 0 8 > /dev/null
[ -f /tmp/wq.bin ] || python tools/dump_workload.py 0.25 /tmp/wq.bin > /dev/null
run() { printf "%-10s %-44s " "$2" "$3"; for rep in 1 2; do timeout 300 .scratch/first_step_probe $1 50 1 60 "$3" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo "ms for 50 steps"; }
for o in "" lane_min_regions=1024 lane_min_regions=1024,lane_width_three=8 lane_min_regions=1024,lane_width_three=4 lane_min_regions=512,lane_width_three=4 ""; do run /tmp/wshard.bin shard "$o"; done
for o in "" lane_width_three=8 lane_width_three=4 ""; do run /tmp/wq.bin quarter "$o"; done
