#!/bin/bash
# Round 4: the large-window leg and the whole-genome step after a change to the wave-per-region kernels (AVK_LIB=libaardvark_amd_prev.so = the build before)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dwfa_scripts.py tests/test_golden_on_kernels.py tests/test_gpu_devpack.py -x -q -m gpu 2>&1 | tail -3
for lib in libaardvark_amd_prev.so libaardvark_amd.so; do echo "== $lib"; AVK_LIB=$lib timeout 600 python tools/gpu_gap_leg.py - 2>&1 | tail -1; done
bash tools/r04_lib_ab.sh libaardvark_amd_prev.so libaardvark_amd.so
