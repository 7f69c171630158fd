"""The 3-caller merge call of bench.py's merge leg in the wide and the packed form; with any argument also the library's stage timing (AVK_TIMING) of one call per form.  usage: python tools/gpu_merge_timing.py [timing]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import aardvark_amd
from aardvark_amd import synth
from aardvark_amd.merge import MergeConfig, merge_multi_batch, pinned_multi_batch, MultiBatch, PackedMultiBatch
ctx = aardvark_amd.Context(0)
for kv in os.environ.get("AVK_OPTS", "").split(","):  # context options, e.g. AVK_OPTS=static_pct=0
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
contigs5, mb = synth.config_genome_merge(scale=1.0, k=3, threads=8)
ctx.upload_reference(contigs5)
print("regions", mb.n_regions, "bytes", sum(getattr(mb, f).nbytes for f in MultiBatch.FIELDS) / 1e6, "MB", {f: getattr(mb, f).nbytes // 1000000 for f in MultiBatch.FIELDS})
mcfg = MergeConfig(majority_voting_enabled=True)
pk = PackedMultiBatch.from_multi(mb)
print("packed form: %.1f MB" % (pk.nbytes() / 1e6))
ref = None
for name, form in (("wide, pinned", pinned_multi_batch(ctx, mb)), ("packed, pinned", pinned_multi_batch(ctx, pk)), ("packed, pageable", pk)):
    for _ in range(3): res = merge_multi_batch(ctx, form, mcfg)
    ref = ref or res
    ts = []
    for _ in range(12):
        t = time.perf_counter(); res = merge_multi_batch(ctx, form, mcfg); ts.append((time.perf_counter() - t) * 1e3)
    print("%-18s %s ms per call; same outputs %s" % (name, " ".join("%.2f" % x for x in ts), np.array_equal(res.status, ref.status) and np.array_equal(res.classification, ref.classification) and np.array_equal(res.members, ref.members)), flush=True)
    if len(sys.argv) > 1:
        os.environ["AVK_TIMING"] = "1"
        for _ in range(4):
            t = time.perf_counter(); merge_multi_batch(ctx, form, mcfg); print("  call with stage timing: %.2f ms, solver launches %.2f ms" % ((time.perf_counter() - t) * 1e3, ctx.last_solver_ms()), flush=True)
        os.environ.pop("AVK_TIMING")
