"""Host-inclusive timing of the C-ABI on the chr20 batch: upload (pack + H2D), resident solve, download (D2H + unpack)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contig, batch = synth.config_chr20_snv()
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", int(os.environ.get("GM", "0")))
ctx.upload_reference([contig])
for it in range(4):
    t0 = time.perf_counter(); rb = ctx.upload(batch); t1 = time.perf_counter()
    ctx.compare_resident(rb, CompareConfig(enable_sequences=False)); ctx.synchronize(); t2 = time.perf_counter()
    got = ctx.download(rb, group_metrics=bool(int(os.environ.get("GM", "0")))); t3 = time.perf_counter()
    rb.free()
    print("upload %.2f ms  solve %.2f ms  download %.2f ms  -> %.2f M regions/s end to end" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, batch.n_regions / (t3 - t0) / 1e6), flush=True)
# the one call of the boundary (avk_compare_batch): for this batch size the one-shot path of avk_stream.inl
import ctypes as C
from aardvark_amd._abi import ResultBatch
cfg = CompareConfig(enable_sequences=False)
res = ResultBatch(batch, sequences=False, group_metrics=False)
cb, ccfg, ro = batch.c_struct(), cfg.c_struct(), res.c_struct()
for it in range(6):
    t0 = time.perf_counter()
    ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
    t1 = time.perf_counter()
    print("avk_compare_batch %.3f ms -> %.2f M regions/s (one-shot path: %s, lanes %d)" % ((t1 - t0) * 1e3, batch.n_regions / (t1 - t0) / 1e6, ctx.last_compare_was_one_shot(),
                                                                                         ctx.last_lane_solved()), flush=True)
