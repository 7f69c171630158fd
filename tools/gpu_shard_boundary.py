"""The boundary call (avk_compare_packed, pinned arrays) on rank 0's hash shard of an N-rank job: ms per call and, with AVK_TIMING=1 in the environment, the library's own
stage lines of the last call.  usage on the GPU box: [AVK_TIMING=1] python tools/gpu_shard_boundary.py [ranks=8] [opt=value,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import aardvark_amd
from aardvark_amd import synth, dist, CompactBatch, PackedBatch
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
contigs, job = synth.config_genome(scale=1.0)
shard = dist.gather_calls(dist.shard_batch(job, 0, ranks))
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
pk = PackedBatch.from_compact(CompactBatch.from_region_batch(shard))
hb, res = ctx.pinned_packed(pk), ctx.pinned_results(pk, packed="only")
for _ in range(5):
    ctx.solve_packed(hb, res=res)
best = 1e9
for rep in range(5):
    t = time.perf_counter()
    for _ in range(20):
        ctx.solve_packed(hb, res=res)
    best = min(best, (time.perf_counter() - t) / 20 * 1e3)
print("shard of %d: %d regions, %d calls: %.3f ms per avk_compare_packed call (best of 5 x 20)" % (ranks, shard.n_regions, shard.n_variants, best), flush=True)
