"""Solver time of the resident SNV batch at several sizes (is the launch chain bound by throughput or by its tail?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for n_truth in (12500, 25000, 50000, 100000, 200000):
    # the same density as the benchmark: contig length scales with the number of calls
    length = int(synth.CHR20_LEN * n_truth / 50000)
    contig, batch = synth.config_chr20_snv(n_truth=n_truth, contig_len=length, n_intervals=max(10, 1000 * n_truth // 50000), n_extra=max(1, n_truth // 100))
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    ms, kms = [], []
    for it in range(40):
        ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
        ctx.synchronize()
        ms.append(ctx.last_solver_ms()); kms.append(ctx.last_kernel_ms())
    ctx.download(rb, group_metrics=False)
    print("%7d regions: solver %.3f ms (first launch %.3f) -> %.1f M regions/s, %.2f ns/region; tiers %s" % (
        batch.n_regions, np.median(ms[5:]), np.median(kms[5:]), batch.n_regions / np.median(ms[5:]) / 1e3, np.median(ms[5:]) * 1e6 / batch.n_regions, ctx.last_tier_counts()), flush=True)
    rb.free()
