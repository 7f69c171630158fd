#!/bin/bash
# Round 4: the whole -m gpu suite, then the default bench run (all legs), on one box
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04_gpu_tests.txt
timeout 1800 python bench.py 2> gpurun_out/r04_bench.log > gpurun_out/r04_bench.json; echo "bench rc $?"; tail -25 gpurun_out/r04_bench.log
