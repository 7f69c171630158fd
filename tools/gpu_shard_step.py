"""The resident step of rank 0's hash shard of an N-rank job (and of chr20) under context options: what bounds strong scaling.
usage on the GPU box: python tools/gpu_shard_step.py [ranks=8] "opt=value,..." ["opt=value,..." ...]   ("-" = defaults)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, dist
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
contigs, job = synth.config_genome(scale=1.0)
shard = dist.gather_calls(dist.shard_batch(job, 0, ranks))
contig20, chr20 = synth.config_chr20_snv()
cfg = CompareConfig(enable_sequences=False)
for o in sys.argv[2:] or ["-"]:
    ctx = aardvark_amd.Context(0)
    ctx.set_option("emit_group_metrics", 0)
    for kv in (o if o != "-" else "").split(","):
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    out = []
    for name, cs, b in (("shard", contigs, shard), ("chr20", [contig20], chr20)):
        ctx.upload_reference(cs)
        rb = ctx.upload(b)
        for _ in range(10):
            ctx.compare_resident(rb, cfg)
        ctx.synchronize()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(100):
                ctx.compare_resident(rb, cfg)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t) * 10)
        ctx.download(rb, group_metrics=False)
        out.append("%s %.3f ms (lanes %d of %d, tiers %s)" % (name, best, ctx.last_lane_solved(), b.n_regions, ctx.last_tier_counts()))
        rb.free()
    ctx.close()
    print("%-60s %s" % (o, "; ".join(out)), flush=True)
