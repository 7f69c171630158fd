#!/usr/bin/env python3
"""prints the last step's kernel timeline from a rocprofv3 rocpd database: tools/show_timeline.py gpurun_out/prof_<tag>/stats/<tag>_results.db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%' order by start"))
last = max(i for i, r in enumerate(rows) if r[0].startswith("avk_tally_reduce"))
prev = max(i for i, r in enumerate(rows[:last]) if r[0].startswith("avk_tally_reduce"))
t0 = rows[prev + 1][4]
for r in rows[prev + 1:last + 1]:
    print("%-30s grid %-8d wg %-4d lds %-7d start %9.1f us  dur %9.1f us" % (r[0][:30], r[1], r[2], r[3], (r[4] - t0) / 1e3, (r[5] - r[4]) / 1e3))
print("step span %.1f us" % ((rows[last][5] - t0) / 1e3))
