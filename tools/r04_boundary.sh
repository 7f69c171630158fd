#!/bin/bash
# Round 4: parity of the packer / result changes on the GPU, then the boundary call's host stage times and device timeline, then a short bench
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_packed_results.py tests/test_gpu_parity.py tests/test_gpu_devpack.py tests/test_gpu_wide.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_b2_tests.txt
AVK_TIMING=1 timeout 300 python3 tools/boundary_once.py 1.0 4 2>&1 | grep -v "avk pool" | tail -12 | tee gpurun_out/r04_boundary_host.txt
bash tools/profile_boundary.sh r04_boundary 2>&1 | grep -v "^W2026" | tail -70
timeout 900 python bench.py --no-secondary --no-cpu-baseline --steps 50 2> gpurun_out/r04_bench_b2.log > gpurun_out/r04_bench_b2.json; echo "bench rc $?"; tail -4 gpurun_out/r04_bench_b2.log
