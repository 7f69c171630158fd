#!/bin/bash
# Round 4: the packed result form on the GPU (parity), the boundary with it, and the job cut into overlapping pieces
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_packed_results.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_packed_tests.txt
timeout 600 python tools/r04_two_ctx.py 1.0 20 2>&1 | tail -12 | tee gpurun_out/r04_two_ctx.txt
timeout 900 python bench.py --no-secondary --no-cpu-baseline --steps 50 2> gpurun_out/r04_bench_packed.log > gpurun_out/r04_bench_packed.json; echo "bench rc $?"; tail -8 gpurun_out/r04_bench_packed.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_packed.json"))
print("value %.1f M/s, %.2f ms; wide results %s; resident %.2f ms" % (d["value"] / 1e6, d["ms_per_step"], d["wide_results"], d["resident"]["ms_per_step"]))
PY
