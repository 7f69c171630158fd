#!/bin/bash
# One rocprofv3 --pmc pass (counters = arguments) over a short bench run; averages per dispatch and launch geometry.  usage on the GPU box: tools/pmc_pass.sh SQ_INSTS_VALU SQ_WAVE_CYCLES
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_pass
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT" -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --resident-steps 3 --no-supervisor > "$OUT/log.txt" 2>&1
echo "rc $?"
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
for d in glob.glob(sys.argv[1] + "/**/p_results.db", recursive=True):
    c = sqlite3.connect(d).cursor()
    vals = {}
    for kname, grid, lds, cname, avg, n in c.execute("select kernel_name, grid_size, lds_block_size, counter_name, avg(value), count(*) from counters_collection "
                                                     "where kernel_name like 'avk_%' group by kernel_name, grid_size, lds_block_size, counter_name"):
        vals.setdefault((kname.split("(")[0], grid, lds), {})[cname] = avg
    for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if not any(x in k[0] for x in ("lane", "quad", "wide", "region_kernel", "pair")):
            continue
        print("%-28s grid=%-8d lds=%-6d " % k + " ".join("%s=%.4g" % (n, x) for n, x in sorted(v.items())))
PY
