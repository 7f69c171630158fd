#!/bin/bash
# Round 3: the whole-genome boundary call and resident step with and without a communicator (RCCL) in the process, under the stream-priority and
# hardware-queue settings (AVK_STREAM_PRIORITY, GPU_MAX_HW_QUEUES).  usage: tools/r03_rccl_queues.sh [full]   (GPU box; full = also the secondary legs)
B="python bench.py --steps 20 --resident-steps 100 --no-cpu-baseline --no-parity"
[ "$1" = "full" ] || B="$B --no-secondary"
run() { echo "== $1"; env $2 timeout 400 $B 2>&1 >/dev/null | grep -E "timed region|resident leg|secondary" | cut -c1-220; }
run "no communicator, default priority" "AVK_STREAM_PRIORITY=default"
run "no communicator, high-priority side streams" "AVK_STREAM_PRIORITY=high"
run "communicator, default priority" "AVK_BENCH_FORCE_DIST=1 AVK_STREAM_PRIORITY=default"
run "communicator, high-priority side streams" "AVK_BENCH_FORCE_DIST=1 AVK_STREAM_PRIORITY=high"
run "communicator, 16 hardware queues, default priority" "AVK_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=16 AVK_STREAM_PRIORITY=default"
run "no communicator, 4 hardware queues, default priority" "GPU_MAX_HW_QUEUES=4 AVK_STREAM_PRIORITY=default"
run "no communicator, 4 hardware queues, high priority" "GPU_MAX_HW_QUEUES=4 AVK_STREAM_PRIORITY=high"
