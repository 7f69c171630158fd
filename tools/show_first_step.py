import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%' order by start"))
first = min(i for i, r in enumerate(rows) if r[0].startswith("avk_tally_reduce"))
t0 = rows[0][4]
for r in rows[:first + 1]:
    if (r[5] - r[4]) / 1e3 > 150 or r[0].startswith("avk_tally"):
        print("%-34s grid %-8d lds %-7d start %9.1f us  dur %9.1f us" % (r[0][:34], r[1], r[3], (r[4] - t0) / 1e3, (r[5] - r[4]) / 1e3))
