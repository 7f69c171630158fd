"""Host time to QUEUE a resident step against the time the GPU needs for it: 100 steps queued without a wait, the clock read behind the last call and behind the
synchronize.  usage on the GPU box: python tools/gpu_enqueue_time.py [ranks=8]   (rank 0's hash shard of that many ranks; 1 = the whole genome)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aardvark_amd
from aardvark_amd import synth, CompareConfig, dist
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
contigs, job = synth.config_genome(scale=1.0)
batch = dist.gather_calls(dist.shard_batch(job, 0, ranks)) if ranks > 1 else job
cfg = CompareConfig(enable_sequences=False)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
for _ in range(10):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.compare_resident(rb, cfg)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print("%d regions: queued 100 steps in %.2f ms (%.3f ms per step of host time), all done after %.2f ms (%.3f ms per step)" % (batch.n_regions, (t1 - t0) * 1e3, (t1 - t0) * 10, (t2 - t0) * 1e3, (t2 - t0) * 10), flush=True)
