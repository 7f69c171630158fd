"""The longest searches of the large-window leg (genome x 0.05, --min-variant-gap 1000) alone: the K regions with the most calls, a wave per region (team_long_windows 0)
against a workgroup per region (1).  The step is as long as its slowest region.  usage: python tools/gpu_team_top.py [K=16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
from aardvark_amd.dist import take_regions, gather_calls
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
contigs, batch = synth.config_genome(scale=0.05, threads=8, gap=1000)
calls = batch.t_cnt.astype(np.int64) + batch.q_cnt
keep = np.sort(np.argsort(-calls, kind="stable")[:K])
sub = gather_calls(take_regions(batch, keep))
print("regions %d, calls %s" % (sub.n_regions, sorted((sub.t_cnt.astype(int) + sub.q_cnt).tolist(), reverse=True)[:8]), flush=True)
cfg = CompareConfig(enable_sequences=False)
ref = None
for m in (0, 1, 2):
    ctx = aardvark_amd.Context(0)
    ctx.set_option("team_long_windows", m)
    ctx.set_option("team_head_regions", 1024)
    for kv in os.environ.get("AVK_OPTS", "").split(","):
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_reference(contigs)
    rb = ctx.upload(sub)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        ctx.compare_resident(rb, cfg)
        ctx.synchronize()
        ts.append(time.perf_counter() - t)
    res = ctx.download(rb, group_metrics=True)
    if ref is None:
        ref = res
    print("team_long_windows %d: %s ms per step, tiers %s, same: %s" % (m, " ".join("%.1f" % (x * 1e3) for x in ts), ctx.last_tier_counts(), res.diff(ref) == []), flush=True)
    rb.free()
    ctx.close()
