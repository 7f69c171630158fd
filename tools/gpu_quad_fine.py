"""Finer split of a quad's clock ticks (profiling builds with -DAVK_QUAD_FINE=1: staging, =2: phasing search; see avk_quad.inl).
usage on the GPU box: AVK_LIB=libaardvark_amd_lanetiming_f1.so python tools/gpu_quad_fine.py 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aardvark_amd
from aardvark_amd import synth, CompareConfig

which = int(sys.argv[1])
contigs, batch = synth.config_genome(scale=1.0)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
ctx.compare_resident(rb, cfg)
ctx.synchronize()
ctx.compare_resident(rb, cfg)
ctx.download(rb, group_metrics=False)
ph = ctx.debug_phase_cycles()
names = {1: ["everything behind the staging", "record loaded and parsed", "reference window", "FULL tables"],
         2: ["everything outside the search loop", "pop", "node restored", "step + alignments (expansion)", "pushes + kept states", "finalize"]}[which]
for base, what in ((0, "three-call class"), (8, "head of the two-call class")):
    quads = max(int(ph[base + 7]), 1)
    tot = max(int(ph[base + 6]), 1)
    print("%s: %d quads, %.0f ticks per quad in claims" % (what, quads, tot / quads))
    for k, nm in enumerate(names):
        print("   %-40s %6.2f %%" % (nm, 100.0 * int(ph[base + k]) / tot))
