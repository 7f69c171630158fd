#!/usr/bin/env python3
"""Prints the launch timeline of the last step recorded in a rocprofv3 --kernel-trace database:
start and end of every solver kernel relative to the first one (shows how the solo launch overlaps the bulk)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute("select name, grid_x, workgroup_x, lds_size, start, end, stream_id from kernels where name like 'avk_%' order by start"))
# a step ends with avk_tally_reduce
steps, cur_step = [], []
for r in rows:
    cur_step.append(r)
    if r[0].startswith("avk_tally_reduce"):
        steps.append(cur_step)
        cur_step = []
for step in steps[-3:]:
    t0 = step[0][4]
    print("step:")
    for name, gx, wx, lds, s, e, sid in step:
        print("  %-26s grid=%-7d wg=%-4d lds=%-7d stream=%s start=%8.1f us end=%8.1f us dur=%8.1f us" % (name[:26], gx, wx, lds, sid, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
