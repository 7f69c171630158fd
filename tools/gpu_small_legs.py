"""The small and the merge legs of bench.py under context options: chr20 resident step and the 3-caller merge call.  usage: python tools/gpu_small_legs.py [opt=value,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
opts = [kv for kv in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if "=" in kv]
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in opts:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
contig, batch = synth.config_chr20_snv()
ctx.upload_reference([contig])
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
for _ in range(20):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
t = time.perf_counter()
for _ in range(400):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
print("%-40s chr20 resident %.3f ms per step, tiers %s, lanes %d" % (",".join(opts) or "defaults", (time.perf_counter() - t) / 400 * 1e3, ctx.last_tier_counts(), ctx.last_lane_solved()), flush=True)
rb.free()
if "--merge" in sys.argv:
    from aardvark_amd.merge import MergeConfig, merge_multi_batch
    contigs5, mb = synth.config_genome_merge(scale=1.0, k=3, threads=8)
    ctx.upload_reference(contigs5)
    mcfg = MergeConfig(majority_voting_enabled=True)
    os.environ.pop("AVK_TIMING", None)
    merge_multi_batch(ctx, mb, mcfg)
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        merge_multi_batch(ctx, mb, mcfg)
        ts.append((time.perf_counter() - t) * 1e3)
    print("%-40s merge of 3 call sets: %s ms per call" % (",".join(opts) or "defaults", " ".join("%.2f" % x for x in ts)), flush=True)
    from aardvark_amd.merge import pinned_multi_batch
    pm = pinned_multi_batch(ctx, mb)
    merge_multi_batch(ctx, pm, mcfg)
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        merge_multi_batch(ctx, pm, mcfg)
        ts.append((time.perf_counter() - t) * 1e3)
    print("%-40s merge, pinned arrays:   %s ms per call" % (",".join(opts) or "defaults", " ".join("%.2f" % x for x in ts)), flush=True)
