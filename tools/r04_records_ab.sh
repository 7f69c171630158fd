#!/bin/bash
# Round 4: the eager record writers beside the lanes' tile records (side stream) against the build before: GPU parity of the packer, then tools/r04_lib_ab.sh, then the shard.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_packed_results.py tests/test_gpu_devpack.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r04_records_ab_tests.txt
bash tools/r04_lib_ab.sh "$@"
for rep in 1 2; do for lib in "$@"; do
  printf "%-34s shard " "$lib"
  AVK_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-parity --no-merge --no-e2e --steps 20 --resident-steps 50 2>&1 | grep -E "secondary (shard_1_of_8|chr20_snv|dense_mix)" | sed 's/.*secondary //; s/, lane share.*//' | tr '\n' ';'; echo
done; done | tee gpurun_out/r04_records_ab_legs.txt
