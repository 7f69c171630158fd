#!/usr/bin/env python3
"""Timeline of the LAST avk_compare_batch call of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/boundary_once.py: every kernel and every
copy with its start relative to the call's first copy.  usage: summarize_boundary.py <results.db> <out.txt>"""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
for r in cur.execute("select name, grid_x, workgroup_x, lds_size, start, end from kernels where name like 'avk_%'"):
    ev.append((r[4], r[5], "kernel %-36s grid=%-8d wg=%-4d lds=%-6d" % (r[0].split("(")[0], r[1], r[2], r[3])))
mc = ["memory_copies"] if "memory_copies" in tables else []
for t in mc:
    for r in cur.execute("select name, size, start, end from %s" % t):
        ev.append((r[2], r[3], "copy   %-36s bytes=%d" % (r[0], r[1])))
ev.sort()
# the last call starts at the last run of large host-to-device copies that follows a device-to-host copy
starts = [i for i, e in enumerate(ev) if "HOST_TO_DEVICE" in e[2].upper() and "bytes=" in e[2] and int(e[2].split("bytes=")[1]) > (1 << 20) and
          (i == 0 or "DEVICE_TO_HOST" in ev[i - 1][2].upper() or "avk_tally" in ev[i - 1][2] or "unpack" in ev[i - 1][2])]
i0 = starts[-1] if starts else 0
t0 = ev[i0][0]
lines = ["# last boundary call of: rocprofv3 --kernel-trace --memory-copy-trace --stats -- python3 tools/boundary_once.py 1.0 3   (whole genome; packed batch form through avk_compare_packed, packed result form, pinned caller arrays)",
         "# start_us and dur_us relative to the call's first host-to-device copy; tables in the database: " + " ".join(mc)]
for s, e, txt in ev[i0:]:
    lines.append("%10.1f %10.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, txt))
lines.append("# span of the call's device work: %.1f us" % ((max(e for _, e, _ in ev[i0:]) - t0) / 1e3))
agg = {}
for s, e, txt in ev[i0:]:
    k = txt.split()[1] if txt.startswith("kernel") else "copy " + txt.split()[1]
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
lines.append("# totals of the call: " + "; ".join("%s x%d %.1f us" % (k, v[0], v[1]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-60:]))
