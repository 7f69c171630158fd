"""Is the large-window leg (--min-variant-gap 1000) bound by its few largest regions?  The resident step of the whole batch against the batch without its largest
0.1 %, 1 % and 5 % of regions (by calls), and the largest ones alone.  python tools/r04_gap_tail.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
from aardvark_amd.dist import take_regions, gather_calls
contigs, batch = synth.config_genome(scale=0.05, threads=8, gap=1000)
cfg = CompareConfig(enable_sequences=False)
calls = batch.t_cnt.astype(np.int64) + batch.q_cnt
order = np.argsort(-calls, kind="stable")
n = batch.n_regions
print("regions %d; calls per region: median %d, 99 %% %d, 99.9 %% %d, max %d; window lengths: median %d, max %d" % (n, np.median(calls), np.percentile(calls, 99), np.percentile(calls, 99.9), calls.max(),
      np.median(batch.end - batch.start), (batch.end - batch.start).max()), flush=True)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
for label, idx in (("whole batch", np.arange(n)), ("without the largest 0.1 %", np.sort(order[n // 1000:])), ("without the largest 1 %", np.sort(order[n // 100:])),
                   ("without the largest 5 %", np.sort(order[n // 20:])), ("the largest 0.1 % alone", np.sort(order[:n // 1000])), ("the largest 1 % alone", np.sort(order[:n // 100]))):
    sub = gather_calls(take_regions(batch, idx))
    rb = ctx.upload(sub)
    ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    ts = []
    for _ in range(2):
        t = time.perf_counter()
        ctx.compare_resident(rb, cfg)
        ctx.synchronize()
        ts.append(time.perf_counter() - t)
    ctx.download(rb, group_metrics=False)
    print("%-28s %6d regions: %s s per step, tiers %s" % (label, sub.n_regions, " ".join("%.3f" % x for x in ts), ctx.last_tier_counts()), flush=True)
    rb.free()
