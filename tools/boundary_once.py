"""One process, a few host-boundary calls on the benchmark genome with pinned caller arrays: the command rocprofv3 traces for profiles/r03_boundary_*.
usage: python tools/boundary_once.py [scale] [calls] [opt=value,...] [form: packed (default) | compact | wide] [results: packed (default) | wide]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[3] if len(sys.argv) > 3 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
from aardvark_amd import CompactBatch, PackedBatch
form = sys.argv[4] if len(sys.argv) > 4 else "packed"
if form == "wide":
    pb, fn = ctx.pinned_batch(batch), ctx.lib.avk_compare_batch
elif form == "compact":
    pb, fn = ctx.pinned_compact(CompactBatch.from_region_batch(batch)), ctx.lib.avk_compare_compact
else:
    pb, fn = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch))), ctx.lib.avk_compare_packed
res = ctx.pinned_results(pb, packed=False if (sys.argv[5] if len(sys.argv) > 5 else "packed") == "wide" else "only")
cb, ccfg, ro = pb.c_struct(), CompareConfig(enable_sequences=False).c_struct(), res.c_struct()
print("form %s: %.0f MB of caller arrays in" % (form, (pb.nbytes() if hasattr(pb, "nbytes") else 0) / 1e6), flush=True)
for k in range(calls + 1):
    t = time.perf_counter()
    ctx._check(fn(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
    print("call %d: %.2f ms (%d regions, solved %d)" % (k, (time.perf_counter() - t) * 1e3, batch.n_regions, int(res.tally[aardvark_amd.TALLY_LEN - 2])), flush=True)
