#!/bin/bash
# Round 4, on the GPU box: the wide kernel's GPU parity tests, then the step's kernel timeline with and without it, then a short bench run with the parity gate.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -15 | tee gpurun_out/r04_wide_tests.txt
g++ -O2 -std=c++17 -I include -o .scratch/first_step_probe tools/first_step_probe.cpp -L aardvark_amd -laardvark_amd -Wl,-rpath,$R/aardvark_amd -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
bash tools/r03_chain.sh "-" "wide_kernel=0" "wide_blocks=256" "wide_blocks=1024" 2>&1 | tee gpurun_out/r04_chain_first.txt
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --resident-steps 100 2>&1 | tail -3 | tee gpurun_out/r04_bench_first.txt
