"""First-contact GPU check: golden regions + the chr20 SNV workload against the oracle, with timings."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aardvark_amd
from aardvark_amd import synth, RegionBatch, CompareConfig
import oracle_lib

lib = oracle_lib.load()
g = json.load(open(os.path.join(ROOT, "tests/golden/waffle_solver.json")))
regions = [{"start": r["start"], "end": r["end"], "truth": r["truth"], "query": r["query"]} for r in g["regions"]]
batch = RegionBatch.from_regions(regions)
ctx = aardvark_amd.Context(0)
for kv in os.environ.get("AVK_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
ctx.upload_reference([g["contig"].encode()])
want = oracle_lib.compare_batch(lib, batch, [g["contig"].encode()], sequences=True)
got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=True))
print("golden diff:", got.diff(want), "status", got.status.tolist(), flush=True)

n_truth = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
t = time.time()
contig, batch = synth.config_chr20_snv(n_truth=n_truth)
print("generated", batch.n_regions, "regions in %.1fs" % (time.time() - t), flush=True)
ctx.upload_reference([contig])
t = time.time()
want = oracle_lib.compare_batch(lib, batch, [contig], sequences=True, threads=os.cpu_count())
t_or = time.time() - t
print("oracle: %.3fs = %.0f regions/s on %d threads" % (t_or, batch.n_regions / t_or, os.cpu_count()), flush=True)
t = time.time()
got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=True))
t_gpu = time.time() - t
print("gpu e2e (upload+kernel+download): %.3fs; kernel %.3f ms; tiers %s" % (t_gpu, ctx.last_kernel_ms(), ctx.last_tier_counts()), flush=True)
d = got.diff(want)
print("chr20 diff:", d, flush=True)
if d:
    n = batch.n_regions
    bad = np.nonzero((got.status != want.status) | (got.ed_h1 != want.ed_h1) | (got.group_metrics.reshape(n, -1) != want.group_metrics.reshape(n, -1)).any(axis=1))[0]
    print("bad regions", bad[:20], len(bad))
rb = ctx.upload(batch)
for it in range(5):
    ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
    ctx.synchronize()
    ms = ctx.last_kernel_ms()
    print("resident run %d: kernel %.3f ms -> %.2f M regions/s" % (it, ms, batch.n_regions / ms / 1e3), flush=True)
res = ctx.download(rb)
want2 = oracle_lib.compare_batch(lib, batch, [contig], sequences=False, threads=os.cpu_count())
print("resident diff:", res.diff(want2))
pc = ctx.debug_phase_cycles()
if pc[7]:
    names = ["stage", "searchA", "searchB", "metrics_setup", "basepair", "record", "region_total"]
    print("phase ticks per region:", {n: round(pc[i] / pc[7], 1) for i, n in enumerate(names)}, "regions", pc[7])
    sub = ["A_setup", "A_pop", "A_finalise", "A_clone", "A_extend(load+store)", "A_push", "ext_copy", "ext_update"]
    print("inside search A:", {n: round(pc[8 + i] / pc[7], 1) for i, n in enumerate(sub)})
