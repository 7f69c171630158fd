import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
contigs, batch = synth.config_genome(scale=0.05, threads=8, gap=1000)
cfg = CompareConfig(enable_sequences=False)
res = {}
for t in (0, int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    ctx = aardvark_amd.Context(0)
    ctx.set_option("team_long_windows", t)
    ctx.set_option("emit_group_metrics", 0)
    ctx.upload_reference(contigs)
    rb = ctx.upload(batch)
    ctx.compare_resident(rb, cfg)
    res[t] = ctx.download(rb, group_metrics=False)
    rb.free(); ctx.close()
a, b = res[0], res[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
for name in ("status", "ed_h1", "ed_h2", "n_optima"):
    d = np.where(getattr(a, name) != getattr(b, name))[0]
    print(name, "differs in", len(d), "regions", d[:10])
vd = np.where((a.var_expected != b.var_expected) | (a.var_observed != b.var_observed) | (a.var_class != b.var_class) | (a.var_zyg != b.var_zyg))[0]
print("calls differing", len(vd))
if len(vd):
    voff = np.concatenate([[0], np.cumsum(batch.t_cnt.astype(np.int64) + batch.q_cnt)])
    regs = np.unique(np.searchsorted(voff, vd, side="right") - 1)
    print("regions with differing calls", len(regs), regs[:10])
    for r in regs[:6]:
        print("  region", r, "N", int(batch.t_cnt[r]) + int(batch.q_cnt[r]), "T", batch.t_cnt[r], "Q", batch.q_cnt[r], "nopt", a.n_optima[r], b.n_optima[r], "status", a.status[r], b.status[r],
              "obs0", a.var_observed[voff[r]:voff[r+1]][:24], "obs1", b.var_observed[voff[r]:voff[r+1]][:24])
