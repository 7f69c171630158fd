"""Back-to-back resident steps without a sync in between (what bench.py times): wall time per step for a few option strings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contig, batch = synth.config_chr20_snv()
for opts in sys.argv[1:] or [""]:
    ctx = aardvark_amd.Context(0)
    for kv in opts.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            if k not in ("null_stream", "step_sync", "full_sync"):
                ctx.set_option(k, int(v))
    ctx.set_option("emit_group_metrics", 0)
    null_stream = "null_stream" in opts
    per_step_sync = "step_sync" in opts
    if null_stream:
        ctx.set_stream(0)
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    cfg = CompareConfig(enable_sequences=False)
    for _ in range(20):
        ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    res = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.compare_resident(rb, cfg)
            if per_step_sync:
                ctx.last_kernel_ms()  # waits for the first launch of this step, like bench.py does
            if "full_sync" in opts:
                ctx.synchronize()  # waits for the end of the step
        ctx.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e3)
    print("%-30s ms per step: %s -> %.1f M regions/s" % (opts, " ".join("%.4f" % x for x in res), batch.n_regions / min(res) / 1e3), flush=True)
    rb.free(); ctx.close()
