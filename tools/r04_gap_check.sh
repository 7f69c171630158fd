#!/bin/bash
# Round 4: the large-window leg after a change — tiers, step, parity against the oracle through bench.py's leg
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 600 python tools/gpu_gap_leg.py - 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_devpack.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --resident-steps 50 --no-cpu-baseline --no-merge --no-e2e 2>&1 | grep -E "secondary|timed region|resident leg" | tee gpurun_out/r04_gap_check.txt
