#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/boundary_pipelined.py: per batch, when its copies in, its packing kernels, its solver
launches, its unpacking and its copies out ran — and what ran beside what.  usage: summarize_pipeline.py <results.db> <out.txt>"""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
ev = []
for name, start, end in cur.execute("select name, start, end from kernels where name like 'avk_%'"):
    n = name.split("(")[0]
    kind = "unpack" if n.startswith("avk_dp_unpack") else ("pack" if n.startswith("avk_dp_") or n.startswith("avk_ps_") else "solve")
    ev.append((start, end, kind, n))
for name, size, start, end in cur.execute("select name, size, start, end from memory_copies"):
    if size >= (1 << 16):
        ev.append((start, end, "h2d" if "HOST_TO_DEVICE" in name.upper() else ("d2h" if "DEVICE_TO_HOST" in name.upper() else "d2d"), "%d bytes" % size))
ev.sort()
# batches: a solve ends with avk_tally_reduce
spans, cur_b = [], {}
def close():
    global cur_b
    if cur_b:
        spans.append(cur_b)
    cur_b = {}
b_idx = 0
phases = []  # (batch, kind, start, end)
solve_end = [e for e in ev if e[3].startswith("avk_tally_reduce")]
bounds = [e[1] for e in solve_end]
def batch_of_solve(t):
    for i, b in enumerate(bounds):
        if t <= b:
            return i
    return len(bounds)
per = {}
for s, e, kind, n in ev:
    if kind in ("solve",):
        b = batch_of_solve(e)
    else:
        b = None
    per.setdefault((kind, b), []).append((s, e))
t0 = ev[0][0]
lines = ["# rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/boundary_pipelined.py 1.0 8: whole genomes back to back, two in flight in ONE context",
         "# every large copy and the span of every batch's solver launches, in time order (us from the first event); `beside` = what else was running during it"]
# merged list of intervals: solve spans per batch, copies individually merged into runs, pack runs, unpack
runs = []
for (kind, b), iv in per.items():
    if kind == "solve":
        runs.append((min(s for s, _ in iv), max(e for _, e in iv), "solve  batch %d (%d launches)" % (b, len(iv))))
for kind in ("h2d", "d2h", "pack", "unpack"):
    iv = sorted(per.get((kind, None), []))
    cur_s = cur_e = None
    cnt = 0
    for s, e in iv:
        if cur_s is not None and s - cur_e < 150_000:  # the same run: gaps under 0.15 ms
            cur_e = max(cur_e, e)
            cnt += 1
        else:
            if cur_s is not None:
                runs.append((cur_s, cur_e, "%-6s (%d)" % (kind, cnt)))
            cur_s, cur_e, cnt = s, e, 1
    if cur_s is not None:
        runs.append((cur_s, cur_e, "%-6s (%d)" % (kind, cnt)))
runs.sort()
for s, e, what in runs:
    beside = sorted(set(w.split()[0] for s2, e2, w in runs if w != what and s2 < e and e2 > s))
    lines.append("%10.1f .. %10.1f  (%8.1f us)  %-28s beside: %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, what, " ".join(beside) or "-"))
sol = sorted(r for r in runs if r[2].startswith("solve"))
if len(sol) > 3:
    gaps = [(sol[i + 1][0] - sol[i][0]) / 1e3 for i in range(1, len(sol) - 1)]
    lines.append("# distance between the starts of consecutive batches' solver launches (steady state): " + " ".join("%.0f" % g for g in gaps) + " us")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-80:]))
