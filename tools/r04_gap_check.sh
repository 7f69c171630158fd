#!/bin/bash
# Round 4: the large-window leg and the whole-genome step after a change to the wave-per-region kernels
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 600 python tools/gpu_gap_leg.py - 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/sweep_options.sh - - 2>&1 | tail -3
