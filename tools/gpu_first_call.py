"""Cost of the first call on a fresh context (workspace allocation) for a few workspace settings: AVK_TIMING=1 prints the allocation time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aardvark_amd, scenarios
from aardvark_amd import CompareConfig
contigs, batch = scenarios.chr20_small(3000)
for opts in sys.argv[1:] or [""]:
    t0 = time.time()
    ctx = aardvark_amd.Context(0)
    for kv in opts.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    ctx.upload_reference(contigs)
    t1 = time.time()
    res = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False))
    t2 = time.time()
    res = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False))
    t3 = time.time()
    print("%-40s context + reference %.3f s, first call %.3f s, second call %.3f s" % (opts, t1 - t0, t2 - t1, t3 - t2), flush=True)
    ctx.close()
