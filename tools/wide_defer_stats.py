"""Why regions leave the wave-cooperative kernel (avk_wide.inl): counts per hand-over site of solve_wide, from the instrumented emulator
(tests/emu/libavk_emu_stats.so, -DAVK_WIDE_STATS).  usage: python tools/wide_defer_stats.py [scenario] [seed] [n]
scenario: clusters | fuzz | genome (the class C regions of the benchmark genome at scale 0.05)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import emu_lib

subprocess.check_call(["make", "-C", emu_lib.EMU_DIR, "libavk_emu_stats.so"], stdout=subprocess.DEVNULL)
emu_lib._lib = None
_orig = C.CDLL


def _load_stats():
    lib = C.CDLL(os.path.join(emu_lib.EMU_DIR, "libavk_emu_stats.so"))
    return lib


SITES = {1: "pre-status / no calls / more than 8 calls on a side", 2: "window + growth + edit bound > 255", 3: "sequence table does not fit the LDS", 4: "flagged reference word",
         5: "call record out of range / ALT not ACGT or over 32 bases", 6: "inexact node reached its turn", 7: "inexact final cost", 8: "more than 32 tied optima",
         9: "queue over 64 entries", 10: "cost over 16 bits", 11: "ids exhausted", 12: "node pool exhausted", 13: "genotype search (hap 1) handed over",
         14: "genotype search (hap 2) handed over", 15: "distance to the reference past the scratch", 16: "per-type alignment past the scratch"}

if __name__ == "__main__":
    scen = sys.argv[1] if len(sys.argv) > 1 else "clusters"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    # the stats build under the name the loader expects
    real = os.path.join(emu_lib.EMU_DIR, "libavk_emu.so")
    C.CDLL = lambda path, *a, **k: _orig(path.replace("libavk_emu.so", "libavk_emu_stats.so"), *a, **k)
    lib = emu_lib.load()
    C.CDLL = _orig
    lib.emu_wide_defer_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    if scen == "genome":
        from aardvark_amd import synth
        contigs, batch = synth.config_genome(scale=float(os.environ.get("SCALE", "0.05")), threads=8)
        got = emu_lib.compare_batch(batch, contigs, threads=8, lane_kernel=True, n_waves=16, group_metrics=False)
    else:
        import scenarios
        import test_wide_parity as twp
        contigs, batch = twp.het_cluster_regions(seed, n) if scen == "clusters" else scenarios.fuzz_regions(seed, n, max_vars=6, related=0.8, span=(40, 220))
        got = emu_lib.compare_batch(batch, contigs, threads=8, lane_kernel=False, class_c_all=True)
    out = (C.c_uint64 * 64)()
    lib.emu_wide_defer_stats(out, 1)
    print("regions %d, wide solved %d, lanes %d, tiers %s" % (batch.n_regions, got.wide_solved, got.lane_solved, got.tier_counts))
    for k in range(64):
        if out[k]:
            print("  site %2d: %6d  %s" % (k, out[k], SITES.get(k, "")))
