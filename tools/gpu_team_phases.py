"""Phase shares of the longest searches of the large-window leg, a wave per region against a workgroup per region: -DAVK_PHASE_TIMING build
(AVK_LIB=libaardvark_amd_phasetiming.so python tools/gpu_team_phases.py [K=16])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "phasetiming" in os.environ.get("AVK_LIB", "")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
from aardvark_amd.dist import take_regions, gather_calls
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
contigs, batch = synth.config_genome(scale=0.05, threads=8, gap=1000)
calls = batch.t_cnt.astype(np.int64) + batch.q_cnt
sub = gather_calls(take_regions(batch, np.sort(np.argsort(-calls, kind="stable")[:K])))
cfg = CompareConfig(enable_sequences=False)
names = ["stage", "searchA", "searchB", "metrics_setup", "basepair", "record", "region_total"]
for m in (0, 1):
    ctx = aardvark_amd.Context(0)
    ctx.set_option("team_long_windows", m)
    ctx.set_option("team_head_regions", 1024)
    ctx.set_option("emit_group_metrics", 0)
    ctx.upload_reference(contigs)
    rb = ctx.upload(sub)
    ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    t = time.perf_counter()
    ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    dt = time.perf_counter() - t
    ctx.download(rb, group_metrics=False)
    pc = ctx.debug_phase_cycles()
    n = max(int(pc[7]), 1)
    print("team_long_windows %d: step %.1f ms; %d regions, ticks per region (owner wave): total %.1f M; %s" % (
        m, dt * 1e3, n, int(pc[6]) / n / 1e6, ", ".join("%s %.1f M" % (nm, int(pc[i]) / n / 1e6) for i, nm in enumerate(names[:6]))), flush=True)
    sub_names = ["A_setup", "A_pop", "A_finalise", "A_clone", "A_extend(load+store)", "A_push", "ext_copy", "ext_update"]
    print("      inside A: " + ", ".join("%s %.1f M" % (nm, int(pc[8 + i]) / n / 1e6) for i, nm in enumerate(sub_names)), flush=True)
    rb.free()
    ctx.close()
