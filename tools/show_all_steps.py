"""every step of a rocprofv3 kernel trace (a step ends with avk_tally_reduce): its span and the launches that lasted longest or started late: tools/show_all_steps.py <results.db> [min_us=300]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rows = list(con.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%' order by start"))
step, t0 = [], None
for r in rows:
    step.append(r)
    if r[0].startswith("avk_tally_reduce"):
        t0 = step[0][4]
        print("step of %d launches, span %.1f us" % (len(step), (r[5] - t0) / 1e3))
        for q in step:
            if (q[5] - q[4]) / 1e3 >= min_us:
                print("    %-34s grid %-9d lds %-7d start %9.1f us  dur %9.1f us" % (q[0][:34], q[1], q[3], (q[4] - t0) / 1e3, (q[5] - q[4]) / 1e3))
        step = []
