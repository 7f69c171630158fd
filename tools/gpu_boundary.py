"""The host boundary on the benchmark genome: avk_compare_batch (host arrays in, host arrays out) with pinned arrays (avk_host_alloc) and with pageable ones,
each checked against the resident path's download.  usage: [AVK_TIMING=1] python tools/gpu_boundary.py [scale] [calls] [opt=value,...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
from aardvark_amd._abi import ResultBatch

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[3] if len(sys.argv) > 3 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
cfg = CompareConfig(enable_sequences=False)
rb = ctx.upload(batch)
ctx.compare_resident(rb, cfg)
want = ctx.download(rb, group_metrics=False)
t0 = time.perf_counter()
for _ in range(10):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
print("resident: %d regions, %.3f ms per step, lanes %d, tiers %s" % (batch.n_regions, (time.perf_counter() - t0) / 10 * 1e3, ctx.last_lane_solved(), ctx.last_tier_counts()), flush=True)
rb.free()
ccfg = cfg.c_struct()
from aardvark_amd import CompactBatch
for name in ("pageable", "pinned"):
    cbt = CompactBatch.from_region_batch(batch)
    if name == "pinned":
        cbt = ctx.pinned_compact(cbt)
    res = ResultBatch(cbt, sequences=False, group_metrics=False) if name == "pageable" else ctx.pinned_results(cbt)
    cc, ro = cbt.c_struct(), res.c_struct()
    ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(cc), C.byref(ccfg), C.byref(ro)))
    ts = []
    for _ in range(calls):
        t = time.perf_counter()
        ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(cc), C.byref(ccfg), C.byref(ro)))
        ts.append((time.perf_counter() - t) * 1e3)
    print("%s compact arrays (%.0f MB in): avk_compare_compact %s ms -> best %.1f M regions/s, mean %.1f; identical to the resident path: %s" %
          (name, cbt.nbytes() / 1e6, " ".join("%.2f" % x for x in ts), batch.n_regions / min(ts) / 1e3, batch.n_regions / np.mean(ts) / 1e3, res.diff(want) == []), flush=True)
# the packed form (avk_packed_batch): offsets implied by order, computed on the device
from aardvark_amd import PackedBatch
for name in ("pageable", "pinned"):
    pk = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
    if name == "pinned":
        pk = ctx.pinned_packed(pk)
    res = ResultBatch(pk, sequences=False, group_metrics=False) if name == "pageable" else ctx.pinned_results(pk)
    pc, ro = pk.c_struct(), res.c_struct()
    ctx._check(ctx.lib.avk_compare_packed(ctx.handle, C.byref(pc), C.byref(ccfg), C.byref(ro)))
    ts = []
    for _ in range(calls):
        t = time.perf_counter()
        ctx._check(ctx.lib.avk_compare_packed(ctx.handle, C.byref(pc), C.byref(ccfg), C.byref(ro)))
        ts.append((time.perf_counter() - t) * 1e3)
    print("%s packed arrays (%.0f MB in): avk_compare_packed %s ms -> best %.1f M regions/s, mean %.1f; identical to the resident path: %s" %
          (name, pk.nbytes() / 1e6, " ".join("%.2f" % x for x in ts), batch.n_regions / min(ts) / 1e3, batch.n_regions / np.mean(ts) / 1e3, res.diff(want) == []), flush=True)
    del pk, res
# the same call with the compact per-region BASEPAIR groups written too (16 B x (1 + call types) per region)
cbt = ctx.pinned_compact(CompactBatch.from_region_batch(batch))
res = ctx.pinned_results(cbt, bp_groups=True)
cc, ro = cbt.c_struct(), res.c_struct()
ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(cc), C.byref(ccfg), C.byref(ro)))
ts = []
for _ in range(calls):
    t = time.perf_counter()
    ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(cc), C.byref(ccfg), C.byref(ro)))
    ts.append((time.perf_counter() - t) * 1e3)
print("pinned compact arrays + BASEPAIR groups (%.0f MB more out): avk_compare_compact %s ms -> best %.1f M regions/s; per-call outputs identical: %s" %
      ((int(res.bp_off[-1]) * 16 + res.bp_off.nbytes) / 1e6, " ".join("%.2f" % x for x in ts), batch.n_regions / min(ts) / 1e3, res.diff(want) == []), flush=True)
del cbt, res
for name in ("pageable", "pinned"):
    b = batch if name == "pageable" else ctx.pinned_batch(batch)
    res = ResultBatch(b, sequences=False, group_metrics=False) if name == "pageable" else ctx.pinned_results(b)
    cb, ro = b.c_struct(), res.c_struct()
    ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
    ts = []
    for _ in range(calls):
        t = time.perf_counter()
        ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
        ts.append((time.perf_counter() - t) * 1e3)
    print("%s arrays: avk_compare_batch %s ms -> best %.1f M regions/s, mean %.1f; identical to the resident path: %s" %
          (name, " ".join("%.2f" % x for x in ts), batch.n_regions / min(ts) / 1e3, batch.n_regions / np.mean(ts) / 1e3, res.diff(want) == []), flush=True)
