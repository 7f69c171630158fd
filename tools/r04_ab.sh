#!/bin/bash
# Round 4: A/B of context options on one box (parity first)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lane.py -x -q -m gpu 2>&1 | tail -2
bash tools/sweep_options.sh "$@" 2>&1 | tee gpurun_out/r04_ab.txt
