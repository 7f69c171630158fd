#!/bin/bash
# Round 4: the kernel timeline of one 3-caller merge call (packed form), and the call under lane thresholds
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
bash tools/profile_merge.sh > /dev/null 2>&1
db=$(ls gpurun_out/prof_merge/*/merge_results.db gpurun_out/prof_merge/merge_results.db 2>/dev/null | head -1)
python3 - "$db" <<'PY' | tee gpurun_out/r04_merge_chain.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%' order by start"))
red = [i for i, r in enumerate(rows) if r[0].startswith("avk_dp_merge_classify") or r[0].startswith("avk_merge_classify")]
print("classify launches:", len(red))
if len(red) >= 2:
    lo, hi = red[-2] + 1, red[-1] + 1
    t0 = rows[lo][4]
    for r in rows[lo:hi]:
        print("%-34s grid=%-9d lds=%-6d start_us=%-9.1f dur_us=%.1f" % (r[0].split("(")[0], r[1], r[3], (r[4] - t0) / 1e3, (r[5] - r[4]) / 1e3))
PY
find gpurun_out/prof_merge -name "*.db" -delete
for o in "" lane_min_regions=8192 lane_min_regions=512; do echo "== ${o:-defaults}"; AVK_OPTS=$o timeout 300 python tools/gpu_merge_timing.py 2>&1 | grep "packed, pinned" | cut -c1-150; done | tee -a gpurun_out/r04_merge_chain.txt
