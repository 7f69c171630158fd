"""The headline call in a fresh process: avk_compare_packed on the benchmark genome (or rank 0's shard), pinned arrays, packed results with BASEPAIR groups.
usage on the GPU box: python tools/gpu_boundary_time.py [ranks=1] [calls=40] [opt=value,...]   (kernel_copies=2: copies by kernel in every process — takes the
half-rate DMA engine that every second process draws out of a comparison)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch, dist
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
contigs, job = synth.config_genome(scale=1.0)
batch = dist.gather_calls(dist.shard_batch(job, 0, ranks)) if ranks > 1 else job
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[3] if len(sys.argv) > 3 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
pb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
res = ctx.pinned_results(pb, packed="only", bp_groups="packed")
cb, ccfg, ro = pb.c_struct(), CompareConfig(enable_sequences=False).c_struct(), res.c_struct()
ts = []
for k in range(calls + 5):
    t = time.perf_counter()
    ctx._check(ctx.lib.avk_compare_packed(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
    ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts[5:])
print("%d regions: boundary call min %.3f median %.3f p90 %.3f ms" % (batch.n_regions, ts.min(), np.median(ts), np.percentile(ts, 90)), flush=True)
