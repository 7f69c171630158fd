#!/bin/bash
# Round 4: the merge call under two in-tree builds of the library on one box (AVK_LIB), alternating, after the GPU parity tests of the merge path.
# usage: tools/r04_merge_ab.sh libaardvark_amd_prev.so libaardvark_amd.so
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_merge.py tests/test_merge_outputs.py tests/test_gpu_lane.py tests/test_gpu_devpack.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r04_merge_ab.txt
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib"; AVK_LIB=$lib timeout 600 python tools/gpu_merge_timing.py timing 2>&1 | grep -E "ms per call|stage timing"
done; done | tee -a gpurun_out/r04_merge_ab.txt
