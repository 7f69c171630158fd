"""A/B of a context option inside one process: blocks of queued whole-genome resident steps and of boundary calls, alternating between two values of the option
(box-to-box differences cancel).  usage: python tools/gpu_static_ab.py [option=static_pct] [a=75] [b=0] [blocks=8]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch
opt = sys.argv[1] if len(sys.argv) > 1 else "static_pct"
va, vb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (75, 0)
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 8
contigs, batch = synth.config_genome(scale=1.0)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
pb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
res = ctx.pinned_results(pb)
cb, ccfg, ro = pb.c_struct(), cfg.c_struct(), res.c_struct()
for _ in range(20):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
step = {va: [], vb: []}
call = {va: [], vb: []}
for k in range(2 * blocks):
    v = va if k % 2 == 0 else vb
    ctx.set_option(opt, v)
    t = time.perf_counter()
    for _ in range(60):
        ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    step[v].append((time.perf_counter() - t) / 60 * 1e3)
    for _ in range(40):
        t = time.perf_counter()
        ctx._check(ctx.lib.avk_compare_packed(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
        call[v].append((time.perf_counter() - t) * 1e3)
for v in (va, vb):
    c = np.array(call[v])
    print("%s=%d: resident step per block %s ms (mean %.3f); boundary calls n=%d mean %.2f median %.2f p95 %.2f max %.2f, above 11 ms: %d" % (
        opt, v, " ".join("%.2f" % x for x in step[v]), np.mean(step[v]), c.size, c.mean(), np.median(c), np.percentile(c, 95), c.max(), int((c > 11).sum())), flush=True)
