#!/usr/bin/env python3
"""Summarises a tools/profile_round.sh run (rocpd databases under gpurun_out/prof_<tag>/) into profiles/<tag>_kernel_stats.txt,
profiles/<tag>_pmc.txt and profiles/<tag>_pmc_traffic.json.  Kernels are reported per launch geometry (name, grid, LDS bytes)."""
import glob, json, os, sqlite3, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --resident-steps 5   (%s): the calls of the three boundary forms (avk_compare_packed timed, avk_compare_compact and avk_compare_batch beside it) and the resident steps, each with one set of solver launches" % tag]
cur = sqlite3.connect(os.path.join(src, "stats", tag + "_results.db")).cursor()
lines.append("# per kernel (all dispatches): name, calls, total_us, avg_us, pct")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append("%-60s calls=%-6d total_us=%-12.1f avg_us=%-10.3f pct=%.2f" % (r[0][:60], r[1], r[2], r[3], r[4]))
lines.append("")
lines.append("# solver kernels per launch geometry")
geo = {}
for r in cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size, count(*), avg(duration), min(duration), max(duration) "
                     "from kernels where name like 'avk_%' group by name, grid_x, lds_size order by avg(duration) desc"):
    lines.append("%-28s grid=%-9d wg=%-4d lds=%-7d vgpr=%-4d sgpr=%-4d scratch=%-4d calls=%-3d avg_us=%-10.1f min_us=%-10.1f max_us=%.1f" %
                 (r[0].split("(")[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8] / 1e3, r[9] / 1e3, r[10] / 1e3))
    geo[(r[0].split("(")[0], r[1], r[3])] = r[8] / 1e3
rows = list(cur.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%' order by start"))
red = [i for i, r in enumerate(rows) if r[0].startswith("avk_tally_reduce")]
if len(red) >= 2:
    lines.append("")
    lines.append("# timeline of the last step")
    t0 = rows[red[-2] + 1][4]
    for r in rows[red[-2] + 1:red[-1] + 1]:
        lines.append("%-28s grid=%-9d lds=%-7d start_us=%-10.1f dur_us=%.1f" % (r[0].split("(")[0], r[1], r[3], (r[4] - t0) / 1e3, (r[5] - r[4]) / 1e3))
    lines.append("step span us = %.1f" % ((rows[red[-1]][5] - t0) / 1e3))
open(os.path.join(dst, tag + "_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
pm = ["# rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 5 --warmup 1 ... (one run per set); averages per dispatch, per launch geometry"]
vals = {}
for d in sorted(glob.glob(os.path.join(src, "pmc*", tag + "_results.db"))):
    c = sqlite3.connect(d).cursor()
    try:
        for kname, grid, lds, cname, avg, n in c.execute("select kernel_name, grid_size, lds_block_size, counter_name, avg(value), count(*) from counters_collection "
                                                         "where kernel_name like 'avk_%' group by kernel_name, grid_size, lds_block_size, counter_name"):
            vals.setdefault((kname.split("(")[0], grid, lds), {})[cname] = avg
    except Exception as e:
        pm.append("%s: %s" % (d, e))
# HBM traffic per STEP: a launch geometry's average per dispatch times its dispatches per step (two hand-back launches of the wide kernel share a geometry), by launch class.
# avk_ps_* (the prefix sums of the packed form) and avk_dp_* are the packing of a boundary call, not the solver's.
steps = {}
for d in sorted(glob.glob(os.path.join(src, "pmc*", tag + "_results.db"))):
    c = sqlite3.connect(d).cursor()
    try:
        for kname, grid, lds, cname, n in c.execute("select kernel_name, grid_size, lds_block_size, counter_name, count(*) from counters_collection where kernel_name like 'avk_%' "
                                                    "group by kernel_name, grid_size, lds_block_size, counter_name"):
            steps.setdefault(cname, {})[(kname.split("(")[0], grid, lds)] = n
    except Exception:
        pass
def klass(name):
    if name.startswith("avk_dp_unpack"):
        return "results (dp_unpack)"
    if name.startswith("avk_dp_") or name.startswith("avk_ps_") or name.startswith("avk_pack"):
        return "packing"
    for key, label in (("avk_quad", "lanes (quads)"), ("avk_lane", "lanes"), ("avk_pair", "looked-up pairs"), ("avk_wide", "wide"), ("avk_region_kernel_lds", "wave-per-region, LDS tiers"),
                       ("avk_region_kernel_hbm", "wave-per-region, HBM tier"), ("avk_tally", "tally reduce")):
        if name.startswith(key):
            return label
    return "other"
per_class = {}
have_traffic = False
for k in sorted(vals, key=lambda k: -vals[k].get("SQ_WAVE_CYCLES", 0)):
    v = vals[k]
    pm.append("")
    pm.append("%s grid=%d lds=%d" % k)
    for name in sorted(v):
        pm.append("    %-24s %.1f" % (name, v[name]))
    if "SQ_WAVE_CYCLES" in v:
        for a in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            if a in v:
                pm.append("    %s / SQ_WAVE_CYCLES = %.3f" % (a, v[a] / v["SQ_WAVE_CYCLES"]))
    if "SQC_ICACHE_REQ" in v and v["SQC_ICACHE_REQ"]:
        pm.append("    instruction cache hit rate = %.4f" % (v.get("SQC_ICACHE_HITS", 0) / v["SQC_ICACHE_REQ"]))
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        t = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        n_red = max(sum(n for kk, n in steps.get("FETCH_SIZE", {}).items() if kk[0].startswith("avk_tally_reduce")), 1)
        per_step = steps.get("FETCH_SIZE", {}).get(k, n_red) / n_red
        pm.append("    HBM bytes per dispatch (FETCH_SIZE + WRITE_SIZE, KiB units) = %.0f; dispatches per step %.2f" % (t, per_step))
        per_class[klass(k[0])] = per_class.get(klass(k[0]), 0.0) + t * per_step
        have_traffic = True
solver = sum(v for c, v in per_class.items() if c not in ("packing", "results (dp_unpack)", "other"))
pm.append("")
pm.append("# HBM bytes per step by launch class (packing and results belong to the boundary call: per call, counted over the steps that ran them)")
for c, v in sorted(per_class.items(), key=lambda kv: -kv[1]):
    pm.append("    %-32s %.0f" % (c, v))
pm.append("    %-32s %.0f" % ("solver launches of a step", solver))
open(os.path.join(dst, tag + "_pmc.txt"), "w").write("\n".join(pm) + "\n")
print("\n".join(pm))
try:  # the build the counters were collected on (tools/profile_round.sh asks the library on the GPU box)
    source_hash = open(os.path.join(src, "source_hash.txt")).read().strip()
except OSError:
    source_hash = None
json.dump({"tag": tag, "source_hash": source_hash, "hbm_bytes_per_launch": solver if have_traffic else None, "hbm_bytes_per_step_by_class": per_class,
           "note": "sum over the SOLVER launches of one step (lanes, looked-up pairs, wide, wave-per-region tiers, tally reduce): (FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch x dispatches per "
                   "step; the packing kernels (avk_dp_*, avk_ps_*) and the result unpacking are listed on their own",
           "per_kernel_avg_us": {"%s grid=%d lds=%d" % k: v for k, v in geo.items()}}, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
