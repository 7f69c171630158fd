"""gpu_time.py for one library with the tier counts of the last run (tuning experiments)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contig, batch = synth.config_chr20_snv()
for opts in sys.argv[1:] or [""]:
    if opts.startswith("lib="):
        lib, _, opts = opts.partition(",")
        os.environ["AVK_LIB"] = lib[4:]
        aardvark_amd.api._lib = None
    ctx = aardvark_amd.Context(0)
    for kv in opts.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    ctx.set_option("emit_group_metrics", 0)
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    ms = []
    for it in range(40):
        ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
        ctx.synchronize()
        ms.append(ctx.last_solver_ms())
    ctx.download(rb, group_metrics=False)
    print("%-50s solver ms median %.3f -> %.1f M regions/s; tiers %s" % (opts, np.median(ms[5:]), batch.n_regions / np.median(ms[5:]) / 1e3, ctx.last_tier_counts()), flush=True)
    rb.free(); ctx.close()
