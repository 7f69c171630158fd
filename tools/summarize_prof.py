#!/usr/bin/env python3
"""Summarises a tools/profile_gpu.sh run (rocprofv3 rocpd databases under gpurun_out/prof_<tag>/)
into profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.txt and profiles/<tag>_pmc_traffic.json."""
import glob
import json
import os
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

lines = []
db = os.path.join(src, "stats", tag + "_results.db")
con = sqlite3.connect(db)
cur = con.cursor()
lines.append("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline   (%s)" % tag)
lines.append("# per kernel (all dispatches): name, calls, total_us, avg_us, pct")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append("%-60s calls=%-6d total_us=%-12.1f avg_us=%-10.3f pct=%.2f" % (r[0][:60], r[1], r[2], r[3], r[4]))
lines.append("")
lines.append("# solver kernel dispatches grouped by launch geometry: avk_region_kernel_lds full grid = dominant first pass (small LDS slices),")
lines.append("# avk_region_kernel_lds with 160 KB LDS = overflow pass with large slices, avk_region_kernel_hbm = HBM tiers (normally empty)")
q = ("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, count(*), avg(duration), min(duration), max(duration) "
     "from kernels where name like 'avk_region%' group by name, grid_x, lds_size order by avg(duration) desc")
main_avg_ns = None
for r in cur.execute(q):
    lines.append("%s grid=%d wg=%d lds=%d vgpr=%d agpr=%d sgpr=%d calls=%d avg_us=%.3f min_us=%.3f max_us=%.3f" %
                 (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8] / 1e3, r[9] / 1e3, r[10] / 1e3))
    if main_avg_ns is None:
        main_avg_ns = r[8]
        main_grid = r[1]
open(os.path.join(dst, tag + "_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# PMC passes: average per main-pass dispatch of the solver kernel
pm = ["# rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity (one run per set)",
      "# averages over the dispatches of the dominant kernel: avk_region_kernel_lds, first pass (the largest grid)"]
vals = {}
for d in sorted(glob.glob(os.path.join(src, "pmc*", tag + "_results.db"))):
    c = sqlite3.connect(d).cursor()
    try:
        g = c.execute("select max(grid_size) from counters_collection where kernel_name like 'avk_region_kernel_lds%'").fetchone()[0]
        for name, avg, n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like 'avk_region_kernel_lds%' and grid_size=? group by counter_name", (g,)):
            vals[name] = avg
            pm.append("%-24s avg_per_launch=%-20.3f launches=%d" % (name, avg, n))
    except Exception as e:
        pm.append("%s: %s" % (d, e))
traffic = None
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    traffic = fetch + write
    pm.append("")
    pm.append("HBM bytes per launch (FETCH_SIZE*1024 + WRITE_SIZE*1024): read %.0f + write %.0f = %.0f" % (fetch, write, traffic))
    pm.append("(gfx950 note, MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide coalesced streams by 2x; this kernel's reads are")
    pm.append(" small per-region gathers, so the raw value is reported and 2x read is the upper bound: %.0f)" % (2 * fetch + write))
if "SQ_WAVE_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
    pm.append("")
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU"):
        if k in vals:
            pm.append("%s / SQ_WAVE_CYCLES = %.3f" % (k, vals[k] / vals["SQ_WAVE_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in vals and "SQ_LDS_IDX_ACTIVE" in vals and vals["SQ_LDS_IDX_ACTIVE"]:
        pm.append("SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f" % (vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]))
if "TCC_HIT_sum" in vals:
    pm.append("L2 hit rate = %.3f" % (vals["TCC_HIT_sum"] / max(1.0, vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])))
open(os.path.join(dst, tag + "_pmc.txt"), "w").write("\n".join(pm) + "\n")
print("\n".join(pm))
json.dump({"tag": tag, "hbm_bytes_per_launch": traffic, "counters": vals, "main_pass_avg_kernel_us": (main_avg_ns or 0) / 1e3},
          open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
