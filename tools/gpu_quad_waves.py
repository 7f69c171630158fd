"""When the waves of the quad launches of one resident step start and end (profiling build: make -C aardvark_amd/csrc quad-wave-log): is a launch as long as its
work divided by the machine, or as long as its last waves?
usage on the GPU box: AVK_LIB=libaardvark_amd_quadwaves.so python tools/gpu_quad_waves.py [scale] [opt=value,...] > gpurun_out/quadwaves.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def describe(log, tick_us=0.01):
    """log[slot][wave] = (t0, t1, claims, longest); s_memrealtime ticks of 10 ns"""
    import statistics
    names = {0: "one call per side, 16 records per wave", 1: "one call per side, narrower", 2: "two calls per side, 16 records per wave", 3: "two calls per side, narrower",
             4: "three calls per side, 16 records per wave", 5: "three calls per side, narrower"}
    for slot in range(6):
        step = [tuple(int(x) for x in w) for w in log[slot] if w[1] > 0]
        if not step:
            continue
        t0 = min(w[0] for w in step)
        end = max(w[1] for w in step)
        dur = (end - t0) * tick_us
        busy = sum(w[1] - w[0] for w in step) * tick_us
        starts = sorted((w[0] - t0) * tick_us for w in step)
        ends = sorted((w[1] - t0) * tick_us for w in step)
        lives = sorted((w[1] - w[0]) * tick_us for w in step)
        print("%s: %d waves, first start to last end %.0f us, sum of wave lifetimes %.0f us = %.0f waves for the whole span" % (names[slot], len(step), dur, busy, busy / max(dur, 1)))
        print("   wave starts: median %.0f us, 90 %% %.0f, last %.0f; wave ends: 10 %% %.0f, median %.0f, 90 %% %.0f, 99 %% %.0f, last %.0f" % (
            statistics.median(starts), starts[int(0.9 * len(starts))], starts[-1], ends[int(0.1 * len(ends))], statistics.median(ends), ends[int(0.9 * len(ends))], ends[int(0.99 * len(ends))], ends[-1]))
        print("   wave lifetime: median %.0f us, 90 %% %.0f, max %.0f; claims per wave: median %d, max %d; longest claim: median %.0f us, 99 %% %.0f, max %.0f" % (
            statistics.median(lives), lives[int(0.9 * len(lives))], lives[-1], statistics.median([w[2] for w in step]), max(w[2] for w in step),
            statistics.median([w[3] for w in step]) * tick_us, sorted(w[3] for w in step)[int(0.99 * len(step))] * tick_us, max(w[3] for w in step) * tick_us))
        print("   waves alive at 10/25/50/75/90 %% of the span: %s" % " ".join(
            str(sum(1 for w in step if (w[0] - t0) * tick_us <= dur * q and (w[1] - t0) * tick_us >= dur * q)) for q in (0.1, 0.25, 0.5, 0.75, 0.9)))


assert "quadwaves" in os.environ.get("AVK_LIB", "")
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
import ctypes as C
import numpy as np
log = np.zeros((6, 4096, 4), np.uint64)
ctx.lib.avk_debug_wave_log.argtypes = [C.c_void_p]
for rep in range(3):
    ctx.compare_resident(rb, cfg)
    ctx.synchronize()
    assert ctx.lib.avk_debug_wave_log(log.ctypes.data) == 0
    print("=== step %d" % rep)
    describe(log)
rb.free()
ctx.close()
