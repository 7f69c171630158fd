#!/bin/bash
# Round 4: two in-tree builds of the library against each other on one box (AVK_LIB), alternating: boundary call and resident step of the whole-genome bench.
# usage: tools/r04_lib_ab.sh libaardvark_amd_prev.so libaardvark_amd.so
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for rep in 1 2; do for lib in "$@"; do
  printf "%-34s " "$lib"
  AVK_LIB=$lib timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-parity --steps 40 --resident-steps 300 2>&1 | grep -E "timed region|resident leg" | grep -v '"metric"' | sed 's/.*(\([0-9.]* ms per call\).*/\1/; s/.*resident leg: /resident /' | tr '\n' ' '; echo
done; done | tee gpurun_out/r04_lib_ab.txt
