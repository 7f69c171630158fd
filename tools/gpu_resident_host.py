"""Where a queued resident step's time goes in a Python host: host time to queue a step against the steps' span on the device, with and without torch in the process
and on the null stream or a stream of the context's own.  usage on the GPU box: python tools/gpu_resident_host.py [torch: 0|1] [null stream: 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
use_torch, null_stream = int(sys.argv[1]), int(sys.argv[2])
if use_torch:
    import torch
    torch.zeros(1, device="cuda:0")
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contigs, batch = synth.config_genome(scale=1.0)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in os.environ.get("AVK_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if null_stream:
    ctx.set_stream(0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
for _ in range(5):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.compare_resident(rb, cfg)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print("torch %d null-stream %d: 100 steps queued in %.2f ms (%.3f ms of host per step), finished after %.2f ms (%.3f ms per step)" % (
        use_torch, null_stream, (t1 - t0) * 1e3, (t1 - t0) * 10, (t2 - t0) * 1e3, (t2 - t0) * 10))
