#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 1200 python -m pytest tests/test_gpu_lane.py tests/test_gpu_parity.py tests/test_gpu_devpack.py tests/test_bench_contract.py tests/test_feeder.py -x -q -m gpu 2>&1 | tail -3
for o in "" lane_min_batch=0 lane_min_batch=16384; do timeout 300 python tools/gpu_small_legs.py "$o" 2>&1 | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1800 python bench.py 2> gpurun_out/r04_bench.log > gpurun_out/r04_bench.json; echo "bench rc $?"; grep -E "timed region|resident leg|secondary" gpurun_out/r04_bench.log
