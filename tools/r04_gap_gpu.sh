#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/r04_gap_tail.py 2>&1 | tail -9 | tee gpurun_out/r04_gap_tail2.txt
