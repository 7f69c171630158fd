"""The large-window leg of bench.py (--min-variant-gap 1000, genome x 0.05) under context options: resident step, tiers, parity of the tally with the default run.
usage: python tools/gpu_gap_leg.py [opt=value,...] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contigs, batch = synth.config_genome(scale=0.05, threads=8, gap=1000)
cfg = CompareConfig(enable_sequences=False)
ref = None
for o in (sys.argv[1:] or ["-"]):
    ctx = aardvark_amd.Context(0)
    ctx.set_option("emit_group_metrics", 0)
    for kv in o.split(","):
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_reference(contigs)
    rb = ctx.upload(batch)
    ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    ts = []
    for _ in range(2):
        t = time.perf_counter()
        ctx.compare_resident(rb, cfg)
        ctx.synchronize()
        ts.append(time.perf_counter() - t)
    res = ctx.download(rb, group_metrics=False)
    if ref is None:
        ref = res
    print("%-40s %d regions: %s s per step, tiers %s, same as the first run: %s" % (o, batch.n_regions, " ".join("%.3f" % x for x in ts), ctx.last_tier_counts(), res.diff(ref) == []), flush=True)
    rb.free()
    ctx.close()
