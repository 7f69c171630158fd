"""Where the wave-cooperative kernel (avk_wide.inl) spends a region's time, whole-genome batch: a -DAVK_WIDE_TIMING build
(make -C aardvark_amd/csrc wide-timing), AVK_LIB=libaardvark_amd_widetiming.so python tools/gpu_wide_phases.py [scale] [opt=value,..]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "widetiming" in os.environ.get("AVK_LIB", "") or "widetrace" in os.environ.get("AVK_LIB", "")
import aardvark_amd
from aardvark_amd import synth, CompareConfig
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs, batch = synth.config_genome(scale=scale)
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
pc = [int(x) for x in ctx.debug_phase_cycles()]
print("tiers", ctx.last_tier_counts(), "lanes", ctx.last_lane_solved(), "wide", ctx.last_wide_solved())
tot, n = max(pc[6], 1), max(pc[7], 1)
print("wide-kernel regions %d (handed over %d), ticks per region %.0f, rounds per region %.1f, pops per region %.1f, entries per round %.1f" %
      (n, pc[11], tot / n, pc[8] / n, pc[9] / n, pc[10] / max(pc[8], 1)))
for i, name in enumerate(["record + tables", "rounds", "commits", "genotype searches", "outputs + alignments", "groups"]):
    print("   %-22s %6.2f %%   %8.0f ticks per region" % (name, 100.0 * pc[i] / tot, pc[i] / n))
print("handed over by: class limits %d, inexact nodes %d, capacities (optima / queue / ids / pool) %d, genotype searches + alignments %d" % tuple(pc[12:16]))
if "widetrace" in os.environ.get("AVK_LIB", ""):
    print("(trace build: the numbers above are the hand-back launch's) waves that took a region %d, mean lifetime %.0f ticks, records written on demand %.0f ticks per region, longest region %d ticks, %d rounds, %d pops, %d calls" % (pc[12], pc[13] / max(pc[12], 1), pc[14] / n, (pc[15] >> 32) << 4, (pc[15] >> 20) & 0xFFF, (pc[15] >> 8) & 0xFFF, pc[15] & 0xFF))
