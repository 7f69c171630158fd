"""Tier counts, parity and solver time of the SNV+indel mix (BASELINE configs[2] shape, one contig)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
n_truth = int(os.environ.get("N_TRUTH", "200000"))
contig, batch = synth.config_indel_mix(n_truth=n_truth)
print("regions", batch.n_regions, "variants", batch.n_variants, flush=True)
for opts in sys.argv[1:] or [""]:
    ctx = aardvark_amd.Context(0)
    for kv in opts.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    ctx.set_option("emit_group_metrics", 0)
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    ms = []
    for it in range(12):
        ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
        ctx.synchronize()
        ms.append(ctx.last_solver_ms())
    got = ctx.download(rb, group_metrics=False)
    print("%-40s solver ms median %.3f -> %.1f M regions/s; tiers %s" % (opts, np.median(ms[2:]), batch.n_regions / np.median(ms[2:]) / 1e3, ctx.last_tier_counts()), flush=True)
    if os.environ.get("CHECK"):
        import oracle_lib
        want = oracle_lib.compare_batch(oracle_lib.load(), batch, [contig], threads=64)
        want.group_metrics = None
        print("diff", got.diff(want), flush=True)
    rb.free(); ctx.close()
