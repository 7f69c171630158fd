"""What every region costs its QUAD (avk_quad.inl), by phase: a -DAVK_LANE_PHASE_TIMING -DAVK_QUAD_REGION_TICKS build writes the region's ticks / 16 into the unused last
group of its metric block.  usage on the GPU box: AVK_LIB=libaardvark_amd_quadticks.so python tools/gpu_quad_regions.py [scale=1.0]
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DAVK_LANE_PHASE_TIMING -DAVK_QUAD_REGION_TICKS -o aardvark_amd/libaardvark_amd_quadticks.so aardvark_amd/csrc/avk_host.hip -ldl)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "quadticks" in os.environ.get("AVK_LIB", "")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
ctx.compare_resident(rb, cfg)
ctx.synchronize()
ctx.compare_resident(rb, cfg)
res = ctx.download(rb, group_metrics=False)
e1, e2 = res.ed_h1.astype(np.int64), res.ed_h2.astype(np.int64)
sel = np.where((e1 & 0x40000000) != 0)[0]
nm = e2[sel] >> 28
ticks = (e1[sel] & 0x3FFFFFFF) * 16
search = (e2[sel] & 0x0FFFFFFF) * 16
print("every region of a tile (16 quads of one wave, in lockstep) carries the TILE's ticks: the numbers below are per tile, weighted by its regions")
for cls, label in ((2, "one call per side"), (4, "two calls per side"), (8, "three calls per side")):
    k = np.where(nm == cls)[0]
    if len(k) == 0:
        continue
    t = ticks[k]
    print("%s: %d regions on quads; tile ticks mean %.0f median %.0f p90 %.0f p99 %.0f p99.9 %.0f max %d (%.0f us at 2.1 GHz); phasing search %.0f %% of the ticks; the slowest 1 %% of regions' tiles hold %.1f %% of the ticks" % (
        label, len(k), t.mean(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), np.percentile(t, 99.9), t.max(), t.max() / 2100.0, 100.0 * search[k].sum() / max(t.sum(), 1),
        100.0 * np.sort(t)[-max(len(t) // 100, 1):].sum() / t.sum()))
    top = k[np.argsort(-t)][:48]
    for r in top:
        o = sel[r]
        def calls(off, cnt):
            return " ".join("%d>%d:z%d" % (batch.a0_len[int(off[o]) + j], batch.a1_len[int(off[o]) + j], batch.var_zyg[int(off[o]) + j]) for j in range(int(cnt[o])))
        print("      region %8d tile ticks %8d search %3.0f %% | L %3d nopt %2d | truth %s | query %s" % (o, ticks[r], 100.0 * search[r] / max(ticks[r], 1), batch.end[o] - batch.start[o], res.n_optima[o],
              calls(batch.t_off, batch.t_cnt), calls(batch.q_off, batch.q_cnt)))
