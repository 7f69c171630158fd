"""Where the lane-per-region kernel spends its time: clock ticks per phase from the profiling build (make -C aardvark_amd/csrc lane-timing).
usage on the GPU box: AVK_LIB=libaardvark_amd_lanetiming.so python tools/gpu_lane_phases.py [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "lanetiming" in os.environ.get("AVK_LIB", ""), "set AVK_LIB=libaardvark_amd_lanetiming.so"
import aardvark_amd
from aardvark_amd import synth, CompareConfig

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in os.environ.get("AVK_OPTS", "").split(","):  # context options by name, e.g. AVK_OPTS=lane_quad=0
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        print("option", kv)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
ctx.compare_resident(rb, cfg)
ctx.synchronize()
ctx.compare_resident(rb, cfg)
ctx.download(rb, group_metrics=False)
ph = ctx.debug_phase_cycles()
names = ["record + window + tables", "search A", "optimum replay + genotypes", "per-call outputs + ed to reference", "per-type alignments", "metric groups + tally",
         "whole tiles"]
for base, what in ((0, "three-call class"), (8, "head of the two-call class")):
    lanes = max(int(ph[base + 7]), 1)
    tot = max(int(ph[base + 6]), 1)
    print("%s: %d lanes, %.0f ticks per lane in tiles" % (what, lanes, tot / lanes))
    for k, nm in enumerate(names[:6]):
        print("   %-40s %6.2f %%" % (nm, 100.0 * int(ph[base + k]) / tot))
    print("   %-40s %6.2f %%" % ("(not in solve_lane: claims, results)", 100.0 * (tot - sum(int(ph[base + k]) for k in range(6))) / tot))
print("lanes solved", ctx.last_lane_solved())
