"""Phase shares of the wave-per-region kernels on the regions the lanes leave them (whole-genome batch): a -DAVK_PHASE_TIMING build
(make -C aardvark_amd/csrc PHASE=1 phase-timing), AVK_LIB=libaardvark_amd_phasetiming.so python tools/gpu_wave_phases.py [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "phasetiming" in os.environ.get("AVK_LIB", "")
import aardvark_amd
from aardvark_amd import synth, CompareConfig
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
gap = int(os.environ.get("GAP", "50"))  # GAP=1000: the large-window workload of bench.py's secondary leg (use scale 0.05)
contigs, batch = synth.config_genome(scale=scale, gap=gap) if gap != 50 else synth.config_genome(scale=scale)
if os.environ.get("LARGEST"):  # LARGEST=0.001: only that share of the regions, the ones with the most calls
    import numpy as np
    from aardvark_amd.dist import take_regions, gather_calls
    calls = batch.t_cnt.astype(np.int64) + batch.q_cnt
    keep = np.sort(np.argsort(-calls, kind="stable")[:max(1, int(batch.n_regions * float(os.environ["LARGEST"])))])
    batch = gather_calls(take_regions(batch, keep))
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
ctx.compare_resident(rb, cfg)
ctx.synchronize()
ctx.compare_resident(rb, cfg)
ctx.download(rb, group_metrics=False)
pc = ctx.debug_phase_cycles()
print("tiers", ctx.last_tier_counts(), "lanes", ctx.last_lane_solved())
names = ["stage", "searchA", "searchB", "metrics_setup", "basepair", "record", "region_total"]
tot = max(int(pc[6]), 1)
print("wave-kernel regions", int(pc[7]), "ticks per region %.0f" % (tot / max(int(pc[7]), 1)))
for i, n in enumerate(names[:6]):
    print("   %-16s %6.2f %%" % (n, 100.0 * int(pc[i]) / tot))
sub = ["A_setup", "A_pop", "A_finalise", "A_clone", "A_extend(load+store)", "A_push", "ext_copy", "ext_update"]
for i, n in enumerate(sub):
    print("   inside A: %-22s %6.2f %%" % (n, 100.0 * int(pc[8 + i]) / tot))
