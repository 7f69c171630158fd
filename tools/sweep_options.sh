#!/bin/bash
# On the GPU box: 50 queued whole-genome resident steps per context option string (one argument each, "-" = defaults; a fresh process each), two repeats.
# Box-to-box differences are +-5 %: compare option strings within one call.  usage: tools/sweep_options.sh - lane_quad=0 wide_kernel=0
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
[ -x .scratch/first_step_probe ] || { mkdir -p .scratch; g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64; }
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-60s " "$1"; for rep in 1 2; do timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo "ms for 50 steps"; }
for o in "$@"; do [ "$o" = "-" ] && o=""; run "$o"; done
