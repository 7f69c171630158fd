#!/bin/bash
# Round 4: the per-rank shard of an 8-GPU run and the large-window workload under context options (fresh process per line, 50 / 3 queued resident steps)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
g++ -O2 -std=c++17 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
[ -f /tmp/wshard.bin ] || python tools/dump_workload.py 1.0 /tmp/wshard.bin 50 0 8 > /dev/null
[ -f /tmp/wgap.bin ] || python tools/dump_workload.py 0.05 /tmp/wgap.bin 1000 > /dev/null
run() { printf "%-64s " "$2 $3"; for rep in 1 2; do timeout 300 .scratch/first_step_probe $1 $4 1 60 "$3" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo "ms for $4 steps"; }
for o in "" lane_min_regions=1024 lane_min_regions=512 lane_min_regions=256 lane_min_regions=0 lane_min_regions=512,wide_blocks=256 lane_min_regions=512,wide_lazy_blocks=128 lane_min_regions=512,hbm_solo_blocks=64 wide_kernel=0 lane_min_regions=512,wide_kernel=0; do run /tmp/wshard.bin shard "$o" 50; done
for o in "" wide_kernel=0 hbm_solo_blocks=256 hbm_solo_blocks=512 waves_per_cu=24; do run /tmp/wgap.bin gap1000 "$o" 3; done
