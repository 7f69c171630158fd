"""Kernel time of the resident chr20 batch for tuning experiments (no parity check)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contig, batch = synth.config_chr20_snv()
for lib in sys.argv[1:]:
    os.environ["AVK_LIB"] = lib
    aardvark_amd.api._lib = None
    ctx = aardvark_amd.Context(0)
    for kv in os.environ.get("AVK_OPTS", "").split(","):
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    ctx.set_option("emit_group_metrics", 0)
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    ms = []
    for it in range(int(os.environ.get("AVK_ITERS", "60"))):
        ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
        ctx.synchronize()
        ms.append(ctx.last_solver_ms())
    print("%-32s kernel ms: median %.3f min %.3f  -> %.1f M regions/s" % (lib, np.median(ms[5:]), min(ms), batch.n_regions / np.median(ms[5:]) / 1e3), flush=True)
    rb.free(); ctx.close()
