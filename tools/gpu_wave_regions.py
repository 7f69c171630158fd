"""How long every region of the wave-per-region kernels takes on its wave (a -DAVK_PHASE_TIMING build writes the region's ticks / 16 and the tier that finished it
into the last two words of its metric block): AVK_LIB=libaardvark_amd_phasetiming.so python tools/gpu_wave_regions.py [scale] [opt=value,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "phasetiming" in os.environ.get("AVK_LIB", "")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
gap = int(os.environ.get("GAP", "50"))  # GAP=1000 with scale 0.05: the large-window leg of bench.py
contigs, batch = synth.config_genome(scale=scale, gap=gap) if gap != 50 else synth.config_genome(scale=scale)
if os.environ.get("SHARD"):  # SHARD=8: rank 0's hash shard of an 8-rank job
    from aardvark_amd import dist
    batch = dist.gather_calls(dist.shard_batch(batch, 0, int(os.environ["SHARD"])))
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 1)
for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
ctx.compare_resident(rb, cfg)
ctx.synchronize()
ctx.compare_resident(rb, cfg)
res = ctx.download(rb, group_metrics=True)
gm = res.group_metrics.reshape(batch.n_regions, -1)
ticks = gm[:, -1].astype(np.int64) * 16
flags = gm[:, -2].astype(np.int64)
tier = flags & 15
kind = flags >> 4  # 1 solo launch, 2 from a list, 4 lazy (handed back by lanes), 8 main HBM launch sharing the class C list
N = batch.t_cnt.astype(np.int64) + batch.q_cnt
L = (batch.end - batch.start).astype(np.int64)
print("tiers", ctx.last_tier_counts(), "lanes", ctx.last_lane_solved())
for t in (1, 2, 3, 4):
    sel = np.where(tier == t)[0]
    if len(sel) == 0:
        continue
    tt = ticks[sel]
    print("tier %d: %d regions, ticks mean %.0f median %.0f p90 %.0f p99 %.0f max %d; sum %.3g" % (t - 1, len(sel), tt.mean(), np.median(tt), np.percentile(tt, 90), np.percentile(tt, 99), tt.max(), tt.sum()))
    o = sel[np.argsort(-tt)][:12]
    for r in o:
        print("     region %d ticks %d N %d (T %d Q %d) L %d nopt %d" % (r, ticks[r], N[r], batch.t_cnt[r], batch.q_cnt[r], L[r], res.n_optima[r]))
    for lo, hi in ((0, 4), (4, 6), (6, 8), (8, 12), (12, 20), (20, 1000)):
        k = sel[(N[sel] >= lo) & (N[sel] < hi)]
        if len(k):
            print("     N in [%d,%d): %d regions, mean ticks %.0f, max %d" % (lo, hi, len(k), ticks[k].mean(), ticks[k].max()))

print("HBM-tier regions by launch:")
names = {1: "HBM solo (class C)", 2 | 8: "main HBM launch: from the overflow list", 8: "main HBM launch: class C leftovers", 2 | 4: "launch for regions the lanes handed back", 2: "list launch"}
for kv in sorted(set(kind[tier == 3].tolist())):
    sel = np.where((tier == 3) & (kind == kv))[0]
    tt = ticks[sel]
    print("   kind %2d %-45s %5d regions, ticks mean %.0f median %.0f p90 %.0f max %d, sum %.3g" % (kv, names.get(kv, "?"), len(sel), tt.mean(), np.median(tt), np.percentile(tt, 90), tt.max(), tt.sum()))
    o = sel[np.argsort(-tt)][:5]
    for r in o:
        print("        region %d ticks %d N %d (T %d Q %d) L %d nopt %d" % (r, ticks[r], N[r], batch.t_cnt[r], batch.q_cnt[r], L[r], res.n_optima[r]))

ph = gm[:, -6:-2].astype(np.int64) * 16  # basepair, metrics_setup, searchB, searchA (words -6 .. -3)
sel = np.where(tier == 3)[0]
o = sel[np.argsort(-ticks[sel])]
def zy(off, cnt, r):
    return "".join(str(int(batch.var_zyg[int(off[r]) + k])) for k in range(int(cnt[r])))
for name, ids in (("the 20 slowest HBM-tier regions", o[:20]), ("ranks 200-210", o[200:210])):
    print(name + ": ticks | searchA searchB setup basepair (%) | N nopt | truth zygosities / query zygosities | allele lengths")
    for r in ids:
        t = max(int(ticks[r]), 1)
        al = " ".join("%d>%d" % (batch.a0_len[int(batch.t_off[r]) + k], batch.a1_len[int(batch.t_off[r]) + k]) for k in range(int(batch.t_cnt[r])))
        print("   %9d | %3.0f %3.0f %3.0f %3.0f | N %2d nopt %2d | %s / %s | %s" % (ticks[r], 100 * ph[r, 3] / t, 100 * ph[r, 2] / t, 100 * ph[r, 1] / t, 100 * ph[r, 0] / t, N[r], res.n_optima[r],
              zy(batch.t_off, batch.t_cnt, r), zy(batch.q_off, batch.q_cnt, r), al))
tot = ph[sel].sum(axis=0)
print("HBM tier, all regions: searchA %.0f %% searchB %.0f %% setup %.0f %% basepair %.0f %%" % tuple(100.0 * tot[[3, 2, 1, 0]] / ticks[sel].sum()))
top = o[:200]
tot = ph[top].sum(axis=0)
print("HBM tier, 200 slowest: searchA %.0f %% searchB %.0f %% setup %.0f %% basepair %.0f %%" % tuple(100.0 * tot[[3, 2, 1, 0]] / ticks[top].sum()))
