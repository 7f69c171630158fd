#!/bin/bash
# Round 4: device timeline of the boundary call (kernels + copies), and the library's own host-side stage times of the same call
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
AVK_TIMING=1 timeout 300 python3 tools/boundary_once.py 1.0 4 2>&1 | tail -30 | tee gpurun_out/r04_boundary_host.txt
bash tools/profile_boundary.sh r04_boundary 2>&1 | tail -90
