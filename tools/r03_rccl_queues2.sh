#!/bin/bash
B="python bench.py --steps 30 --resident-steps 100 --no-cpu-baseline --no-parity --no-secondary"
run() { echo "== $1"; env $2 timeout 400 $B 2>&1 >/dev/null | grep -E "timed region|resident leg" | cut -c1-220; }
run "no communicator, 16 queues, default priority" "GPU_MAX_HW_QUEUES=16 AVK_STREAM_PRIORITY=default"
run "no communicator, 8 queues, default priority" "GPU_MAX_HW_QUEUES=8 AVK_STREAM_PRIORITY=default"
run "no communicator, 16 queues, high priority" "GPU_MAX_HW_QUEUES=16 AVK_STREAM_PRIORITY=high"
run "communicator, 16 queues, default priority" "AVK_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=16 AVK_STREAM_PRIORITY=default"
run "communicator, 24 queues, default priority" "AVK_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=24 AVK_STREAM_PRIORITY=default"
run "communicator, 12 queues, default priority" "AVK_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=12 AVK_STREAM_PRIORITY=default"
run "no communicator, 24 queues, default priority" "GPU_MAX_HW_QUEUES=24 AVK_STREAM_PRIORITY=default"
