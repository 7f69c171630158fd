"""The library's host-side stage times (AVK_TIMING) of boundary calls on rank 0's shard of an 8-rank job (packed batch, packed results, pinned arrays).
python tools/r04_shard_host.py [world] [opt=value,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AVK_TIMING"] = "1"
import ctypes as C
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch
from aardvark_amd.dist import shard_batch, gather_calls
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
contigs, batch = synth.config_genome(scale=1.0)
sub = gather_calls(shard_batch(batch, 0, world))
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
hb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(sub)))
res = ctx.pinned_results(hb, packed="only")
cb, ccfg, ro = hb.c_struct(), CompareConfig(enable_sequences=False).c_struct(), res.c_struct()
for k in range(10):
    t = time.perf_counter()
    ctx._check(ctx.lib.avk_compare_packed(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
    print("call %d: %.3f ms (%d regions)" % (k, (time.perf_counter() - t) * 1e3, sub.n_regions), flush=True)
