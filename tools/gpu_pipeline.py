"""A whole-genome job as K sub-batches (contiguous runs of regions, each with its own slice of the call arrays) through TWO contexts on one GPU, one host
thread each: the copies of one sub-batch run beside the packing and the solve of the other.  Host arrays in (pinned, compact form), host arrays out.
usage: python tools/gpu_pipeline.py [scale] [K ...]"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 6, 8, 12]
contigs, batch = synth.config_genome(scale=scale)
full = CompactBatch.from_region_batch(batch)
ctxs = [aardvark_amd.Context(0), aardvark_amd.Context(0)]
for c in ctxs:
    c.set_option("emit_group_metrics", 0)
    c.upload_reference(contigs)
cfg = CompareConfig(enable_sequences=False).c_struct()


def chunk(cb, r0, r1):
    v0 = int(cb.v_off[r0])
    v1 = int(cb.v_off[r1 - 1]) + int(cb.t_cnt[r1 - 1]) + int(cb.q_cnt[r1 - 1])
    a0 = int(cb.a_off[v0])
    a1 = int(cb.a_off[v1 - 1]) + int(cb.a0_len[v1 - 1]) + int(cb.a1_len[v1 - 1])
    return CompactBatch(contig_idx=cb.contig_idx[r0:r1], start=cb.start[r0:r1], len=cb.len[r0:r1], v_off=cb.v_off[r0:r1] - v0, t_cnt=cb.t_cnt[r0:r1], q_cnt=cb.q_cnt[r0:r1],
                        var_pos=cb.var_pos[v0:v1], var_type_zyg=cb.var_type_zyg[v0:v1], a_off=cb.a_off[v0:v1] - a0, a0_len=cb.a0_len[v0:v1], a1_len=cb.a1_len[v0:v1],
                        var_raw_space=None if cb.var_raw_space is None else cb.var_raw_space[v0:v1], allele_bytes=cb.allele_bytes[a0:a1])


# reference result: one call
c0 = ctxs[0]
pf = c0.pinned_compact(full)
rf = c0.pinned_results(pf)
cc, ro = pf.c_struct(), rf.c_struct()
for _ in range(2):
    c0._check(c0.lib.avk_compare_compact(c0.handle, C.byref(cc), C.byref(cfg), C.byref(ro)))
ts = []
for _ in range(5):
    t = time.perf_counter()
    c0._check(c0.lib.avk_compare_compact(c0.handle, C.byref(cc), C.byref(cfg), C.byref(ro)))
    ts.append((time.perf_counter() - t) * 1e3)
print("one call: %s ms -> %.1f M regions/s" % (" ".join("%.2f" % x for x in ts), full.n_regions / min(ts) / 1e3), flush=True)
want_status, want_ed1, want_tally = rf.status.copy(), rf.ed_h1.copy(), rf.tally.copy()

for K in Ks:
    edges = [full.n_regions * k // K for k in range(K + 1)]
    jobs = []
    for k in range(K):
        ctx = ctxs[k % 2]
        cb = ctx.pinned_compact(chunk(full, edges[k], edges[k + 1]))
        res = ctx.pinned_results(cb)
        jobs.append((ctx, cb, res, cb.c_struct(), res.c_struct()))

    def worker(w):
        for k in range(w, K, 2):
            ctx, cb, res, cs, rs = jobs[k]
            ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(cs), C.byref(cfg), C.byref(rs)))

    def run():
        th = [threading.Thread(target=worker, args=(w,)) for w in range(min(2, K))]
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return (time.perf_counter() - t) * 1e3

    run(); run()
    ts = [run() for _ in range(6)]
    st = np.concatenate([j[2].status for j in jobs])
    ed = np.concatenate([j[2].ed_h1 for j in jobs])
    tl = sum(j[2].tally.astype(np.uint64) for j in jobs)
    same = np.array_equal(st, want_status) and np.array_equal(ed, want_ed1) and np.array_equal(tl, want_tally)
    print("K %2d sub-batches, 2 contexts: %s ms -> %.1f M regions/s; same results as one call: %s" % (K, " ".join("%.2f" % x for x in ts), full.n_regions / min(ts) / 1e3, same), flush=True)
