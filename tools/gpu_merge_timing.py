"""Stage timing (AVK_TIMING) of the 3-caller merge call, bench.py's merge leg, from pinned arrays.  usage: python tools/gpu_merge_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import aardvark_amd
from aardvark_amd import synth
from aardvark_amd.merge import MergeConfig, merge_multi_batch, pinned_multi_batch, MultiBatch
ctx = aardvark_amd.Context(0)
contigs5, mb = synth.config_genome_merge(scale=1.0, k=3, threads=8)
ctx.upload_reference(contigs5)
print("regions", mb.n_regions, "bytes", sum(getattr(mb, f).nbytes for f in MultiBatch.FIELDS) / 1e6, "MB", {f: getattr(mb, f).nbytes // 1000000 for f in MultiBatch.FIELDS})
pm = pinned_multi_batch(ctx, mb)
mcfg = MergeConfig(majority_voting_enabled=True)
for _ in range(3): merge_multi_batch(ctx, pm, mcfg)
os.environ["AVK_TIMING"] = "1"
for _ in range(3):
    t = time.perf_counter(); merge_multi_batch(ctx, pm, mcfg); print("call %.2f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
