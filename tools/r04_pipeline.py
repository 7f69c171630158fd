"""A job of many whole-genome batches through TWO contexts on one GPU, each on a thread of its own, batch i on context i mod 2: the copies of one batch run under the
kernels of the other.  Same calls as bench.py's value leg (avk_compare_packed, packed results, pinned arrays); ms per batch over the job.
python tools/r04_pipeline.py [scale] [batches]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
contigs, batch = synth.config_genome(scale=scale)
ccfg = CompareConfig(enable_sequences=False).c_struct()


def make(k):
    out = []
    for _ in range(k):
        ctx = aardvark_amd.Context(0)
        ctx.set_option("emit_group_metrics", 0)
        ctx.upload_reference(contigs)
        hb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
        res = ctx.pinned_results(hb, packed="only")
        out.append((ctx, hb, res, hb.c_struct(), res.c_struct()))
    return out


def job(workers, n):
    def run(w, count):
        ctx, hb, res, cb, ro = w
        for _ in range(count):
            ctx._check(ctx.lib.avk_compare_packed(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
    ths = [threading.Thread(target=run, args=(w, n // len(workers))) for w in workers]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return (time.perf_counter() - t0) / (n // len(workers) * len(workers)) * 1e3


for k in (1, 2, 1, 2):
    ws = make(k)
    job(ws, 2 * k)
    ms = job(ws, n_batches)
    same = all(np.array_equal(w[2].region_packed, ws[0][2].region_packed) for w in ws)
    print("%d context(s): %.2f ms per whole-genome batch (%.0f M regions/s), results of the contexts identical: %s" % (k, ms, batch.n_regions / ms / 1e3, same), flush=True)
    for w in ws:
        w[0].close()
