"""The team launch (avk_region_kernel_team) on a small batch of large windows under team_long_windows = 0 (a wave per region), 2 (a workgroup per region, the owner takes
every job), 1 (the siblings take jobs too): same outputs, step times.  usage: timeout 120 python tools/gpu_team_probe.py [modes] [contig_len]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,2,1").split(",")]
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
contig, bed, truth, query = synth.contig_calls(5, length, 3_800 / 3_000_000 * length / length * length / 1, seed_ref=905, seed_query=906, str_frac=0.15, multi_frac=0.05) if False else \
    synth.contig_calls(5, length, 3_800 / 3_000_000, seed_ref=905, seed_query=906, str_frac=0.15, multi_frac=0.05)
batch = synth.cluster_regions_v(contig, bed, truth, query, 1000)
print("regions", batch.n_regions, "max calls", int((batch.t_cnt.astype(np.int64) + batch.q_cnt).max()), flush=True)
cfg = CompareConfig(enable_sequences=False)
ref = None
for m in modes:
    ctx = aardvark_amd.Context(0)
    ctx.set_option("team_long_windows", m)
    for kv in os.environ.get("AVK_OPTS", "").split(","):
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    print("mode", m, "launching", flush=True)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        ctx.compare_resident(rb, cfg)
        ctx.synchronize()
        ts.append(time.perf_counter() - t)
    res = ctx.download(rb, group_metrics=True)
    if ref is None:
        ref = res
    print("mode %d: %s ms per step, tiers %s, same as the first mode: %s" % (m, " ".join("%.2f" % (x * 1e3) for x in ts), ctx.last_tier_counts(), res.diff(ref) == []), flush=True)
    rb.free()
    ctx.close()
