"""The boundary call (avk_compare_packed, pinned arrays) on the large-window batch of bench.py (--min-variant-gap 1000, genome x 0.05): ms per call; AVK_TIMING=1 for the stage lines."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch
contigs, batch = synth.config_genome(scale=0.05, threads=8, gap=1000)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
pb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
res = ctx.pinned_results(pb, packed="only")
for k in range(4):
    t = time.perf_counter()
    ctx.solve_packed(pb, res=res)
    print("call %d: %.1f ms" % (k, (time.perf_counter() - t) * 1e3), flush=True)
