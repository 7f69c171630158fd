#!/bin/bash
# Round 4: head launches of the lane classes sized by how many regions they hold (lane_head_auto) against the fixed 16 records per wave, three batch sizes
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 900 python -m pytest tests/test_gpu_lane.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
[ -f /tmp/wshard.bin ] || python tools/dump_workload.py 1.0 /tmp/wshard.bin 50 0 8 > /dev/null
[ -f /tmp/wq.bin ] || python tools/dump_workload.py 0.25 /tmp/wq.bin > /dev/null
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-10s %-24s " "$2" "$3"; for rep in 1 2; do timeout 300 .scratch/first_step_probe $1 50 1 60 "$3" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo "ms for 50 steps"; }
for w in shard:/tmp/wshard.bin quarter:/tmp/wq.bin whole:/tmp/w100.bin; do for o in "" lane_head_auto=0 "" lane_head_auto=0; do run ${w#*:} ${w%%:*} "$o"; done; done
