"""The asynchronous boundary on the whole-genome job: ms per genome-equivalent of (a) the synchronous call, (b) the genome as 2 / 4 / 8 contig-aligned batches in flight
inside one context, (c) genomes back to back, submit(k + 1) before wait(k).  usage on the GPU box: python tools/gpu_async_pipeline.py [scale] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import CompactBatch, PackedBatch, ResultBatch, synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in os.environ.get("AVK_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
whole = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
hb = ctx.pinned_packed(whole)
res = ctx.pinned_results(hb, packed="only")
for _ in range(3):
    ctx.solve_packed(hb, res=res)
t0 = time.perf_counter()
for _ in range(reps):
    ctx.solve_packed(hb, res=res)
sync_ms = (time.perf_counter() - t0) / reps * 1e3
ref_rp, ref_vp, ref_tally = res.region_packed.copy(), res.var_packed.copy(), res.tally.copy()
print("synchronous avk_compare_packed: %.3f ms per genome (%d regions)" % (sync_ms, batch.n_regions))
for n_parts in (2, 4, 8):
    parts = [ctx.pinned_packed(p) for p in whole.split(n_parts)]
    outs = [ctx.pinned_results(p, packed="only") for p in parts]
    def genome():
        tickets = []
        for p, o in zip(parts, outs):
            if len(tickets) == 4:
                tickets.pop(0).wait()
            tickets.append(ctx.submit_packed(p, res=o))
        for t in tickets:
            t.wait()
    genome()
    t0 = time.perf_counter()
    for _ in range(reps):
        genome()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ok = (np.array_equal(np.concatenate([o.region_packed[:p.n_regions] for o, p in zip(outs, parts)]), ref_rp[:whole.n_regions]) and
          np.array_equal(np.concatenate([o.var_packed[:p.n_variants] for o, p in zip(outs, parts)]), ref_vp[:whole.n_variants]) and
          np.array_equal(sum(o.tally.astype(np.uint64) for o in outs), ref_tally))
    print("the genome as %d batches in flight: %.3f ms per genome, outputs identical to the synchronous call: %s" % (len(parts), ms, ok))
# genomes back to back: `depth` sets of arrays in flight, submit(k + depth - 1) before wait(k)
for depth, prefetch in ((2, False), (3, False)):
    nset = depth + (1 if prefetch else 0)
    sets = [(hb, res)] + [(ctx.pinned_packed(whole), ctx.pinned_results(hb, packed="only")) for _ in range(nset - 1)]
    tickets = [ctx.submit_packed(sets[k][0], res=sets[k][1]) for k in range(depth - 1)]
    t0 = time.perf_counter()
    for k in range(depth - 1, reps + depth - 1):
        tickets.append(ctx.submit_packed(sets[k % nset][0], res=sets[k % nset][1]))
        tickets.pop(0).wait()
    for t in tickets:
        t.wait()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ok = all(np.array_equal(r.region_packed, ref_rp) and np.array_equal(r.var_packed, ref_vp) and np.array_equal(r.tally, ref_tally) for _, r in sets)
    print("genomes back to back, %d in flight%s: %.3f ms per genome (%.1f M regions/s), outputs identical: %s" % (depth, " + the next one's copies started ahead" if prefetch else "", ms,
                                                                                                                    batch.n_regions / ms / 1e3, ok))
ctx.close()
