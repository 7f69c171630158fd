#!/usr/bin/env python3
"""Per launch of a step (by position in the step): min / median / max of start offset and duration over all recorded steps."""
import sqlite3, sys
import numpy as np
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%' order by start"))
steps, cs = [], []
for r in rows:
    cs.append(r)
    if r[0].startswith("avk_tally_reduce"):
        steps.append(cs); cs = []
steps = [s for s in steps if len(s) == len(steps[-1])][2:]
print("%d steps of %d launches" % (len(steps), len(steps[-1])))
for k in range(len(steps[-1])):
    st = np.array([(s[k][4] - s[0][4]) / 1e3 for s in steps]); du = np.array([(s[k][5] - s[k][4]) / 1e3 for s in steps])
    n = steps[-1][k]
    print("  %-24s grid=%-7d wg=%-4d start %8.1f/%8.1f/%8.1f us   dur %8.1f/%8.1f/%8.1f us" % (n[0][:24], n[1], n[2], st.min(), np.median(st), st.max(), du.min(), np.median(du), du.max()))
tot = np.array([(s[-1][5] - s[0][4]) / 1e3 for s in steps])
print("  step total %.1f/%.1f/%.1f us" % (tot.min(), np.median(tot), tot.max()))
