"""Would the boundary call gain from being cut in pieces that overlap each other?  The whole-genome job as ONE avk_compare_packed call on one context, against the
same job cut into k contiguous pieces (k = 2, 3, 4), each piece handed to a context of its own on the same GPU by a thread of its own: piece i's copies and
packing kernels can then run under piece j's solver launches.  Packed batch form, packed result form, pinned arrays.
python tools/r04_two_ctx.py [scale] [reps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch
from aardvark_amd.dist import take_regions, gather_calls

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
contigs, batch = synth.config_genome(scale=scale)
n = batch.n_regions
ccfg = CompareConfig(enable_sequences=False).c_struct()


def make_ctx():
    ctx = aardvark_amd.Context(0)
    ctx.set_option("emit_group_metrics", 0)
    ctx.upload_reference(contigs)
    return ctx


def prepare(ctx, sub):
    hb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(sub)))
    res = ctx.pinned_results(hb, packed="only")
    return hb, res, hb.c_struct(), res.c_struct()


def run(k):
    ctxs = [make_ctx() for _ in range(k)]
    cuts = [n * i // k for i in range(k + 1)]
    parts = []
    for i in range(k):
        sub = batch if k == 1 else gather_calls(take_regions(batch, np.arange(cuts[i], cuts[i + 1])))
        parts.append(prepare(ctxs[i], sub))

    def call(i):
        hb, res, cb, ro = parts[i]
        ctxs[i]._check(ctxs[i].lib.avk_compare_packed(ctxs[i].handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))

    def job(stagger):
        if k == 1:
            call(0)
            return
        ths = [threading.Thread(target=call, args=(i,)) for i in range(k)]
        for i, t in enumerate(ths):
            t.start()
            if stagger and i + 1 < k:
                time.sleep(stagger)
        for t in ths:
            t.join()

    out = []
    for stagger in ((0.0,) if k == 1 else (0.0, 0.0005, 0.001)):
        for _ in range(3):
            job(stagger)
        t0 = time.perf_counter()
        for _ in range(reps):
            job(stagger)
        out.append((stagger, (time.perf_counter() - t0) / reps * 1e3))
    tally = sum(p[1].tally.astype(np.uint64) for p in parts)
    for c in ctxs:
        c.close()
    return out, tally


base = None
for k in (1, 2, 3, 4):
    out, tally = run(k)
    if base is None:
        base = tally
    assert np.array_equal(tally, base), "tallies differ"
    for stagger, ms in out:
        print("%d piece(s), threads started %.1f ms apart: %.2f ms per whole job (%.1f M regions/s)" % (k, stagger * 1e3, ms, n / ms / 1e3), flush=True)
