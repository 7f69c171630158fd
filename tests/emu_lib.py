"""Loader for the kernel-logic emulator (tests/emu/libavk_emu.so): the HIP solver source run
on CPU lanes.  Test infrastructure for the GPU-less container."""
import ctypes as C
import os
import subprocess

from aardvark_amd._abi import AvkCompareConfig, AvkRegionBatch, AvkResultBatch, ResultBatch
from oracle_lib import ContigSet, u8p, u64p

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
_lib = None


def load():
    global _lib
    if _lib is None:
        import fcntl
        with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:  # pytest-xdist workers: one of them builds, the others wait (a half-written .so is an OSError)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", EMU_DIR, "libavk_emu.so"], stdout=subprocess.DEVNULL)
        lib = C.CDLL(os.path.join(EMU_DIR, "libavk_emu.so"))
        lib.emu_compare_batch.argtypes = [C.POINTER(AvkRegionBatch), C.POINTER(u8p), u64p, C.c_uint32, C.POINTER(AvkCompareConfig),
                                          C.POINTER(AvkResultBatch), C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64,
                                          C.c_uint32, C.c_int, u64p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.emu_optimize_pairs_batch.argtypes = [C.POINTER(AvkRegionBatch), C.POINTER(u8p), u64p, C.c_uint32, C.c_uint32,
                                                 C.POINTER(C.c_int32), u8p, C.c_int]
        lib.emu_set_lane_kernel.argtypes = [C.c_int]
        lib.emu_last_lane_solved.restype = C.c_uint64
        lib.emu_last_wide_solved.restype = C.c_uint64
        lib.emu_set_wide_kernel.argtypes = [C.c_int]
        lib.emu_set_wide_lds_bytes.argtypes = [C.c_uint32]
        lib.emu_set_lane_pool.argtypes = [C.c_int]
        lib.emu_set_lane_quad.argtypes = [C.c_int]
        lib.emu_last_quad_solved.restype = C.c_uint64
        _lib = lib
    return _lib


def compare_batch(batch, contigs, max_branch_factor=50, sequences=False, exact_shortcut=False,
                  lds_bytes=10 * 1024, lds_ed_cap=48, lds2_bytes=40 * 1024, lds2_ed_cap=48, ws_bytes=1 << 20, big_ws_bytes=64 << 20,
                  n_waves=8, threads=8, solo_min_variants=5, lds2_overflow_pass=0, lds_escalation=1, group_metrics=True, lane_kernel=True, bp_groups=False,
                  wide_kernel=True, wide_lds_bytes=16 * 1024, class_c_all=False, packed=False, lane_pool=-1, lane_quad=True):
    """lane_kernel: small regions go through the lane-per-region code (avk_lane.inl), the rest through the wave-per-region code, as
    avk_compare_resident does; False = everything through the wave-per-region code.  res.lane_solved = regions the lane code finished.
    wide_kernel: class C and what the three-call lane class hands back go through the wave-cooperative code of avk_wide.inl first
    (res.wide_solved = regions it finished); class_c_all = every region outside the lane classes is planned as class C."""
    lib = load()
    lib.emu_set_lane_kernel(1 if lane_kernel else 0)
    lib.emu_set_wide_kernel(1 if wide_kernel else 0)
    lib.emu_set_wide_lds_bytes(wide_lds_bytes)
    lib.emu_set_lane_quad(1 if lane_quad else 0)  # context option lane_quad: launches of at most 16 records per wave run four lanes per region (avk_quad.inl); res.quad_solved
    lib.emu_set_lane_pool(lane_pool)  # context option lane_pool: node states a lane keeps during its search (-1: by class, in the heads and the three-call class)
    before = os.environ.get("AVK_EMU_CLASS_C")
    if class_c_all:
        os.environ["AVK_EMU_CLASS_C"] = "100000"
    cs = contigs if isinstance(contigs, ContigSet) else ContigSet(contigs)
    res = ResultBatch(batch, sequences=sequences, group_metrics=group_metrics, bp_groups=bp_groups, packed=packed)
    cfg = AvkCompareConfig(max_branch_factor, 1 if sequences else 0, 1 if exact_shortcut else 0)
    cb, ro = batch.c_struct(), res.c_struct()
    tiers = (C.c_uint64 * 5)()
    rc = lib.emu_compare_batch(C.byref(cb), cs.ptrs, cs.lens, cs.n, C.byref(cfg), C.byref(ro), lds_bytes, lds_ed_cap, lds2_bytes, lds2_ed_cap,
                               ws_bytes, big_ws_bytes, n_waves, threads, tiers, solo_min_variants, lds2_overflow_pass, lds_escalation)
    assert rc == 0
    res.tier_counts = [int(x) for x in tiers]
    res.lane_solved = int(lib.emu_last_lane_solved())
    res.wide_solved = int(lib.emu_last_wide_solved())
    res.quad_solved = int(lib.emu_last_quad_solved())
    if class_c_all:
        if before is None:
            os.environ.pop("AVK_EMU_CLASS_C", None)
        else:
            os.environ["AVK_EMU_CLASS_C"] = before
    return res


def optimize_pairs(batch, contigs, max_branch_factor=50, threads=8):
    import numpy as np
    lib = load()
    cs = contigs if isinstance(contigs, ContigSet) else ContigSet(contigs)
    status = np.full(batch.n_regions, -1, np.int32)
    exact = np.zeros(max(batch.n_regions, 1), np.uint8)
    cb = batch.c_struct()
    rc = lib.emu_optimize_pairs_batch(C.byref(cb), cs.ptrs, cs.lens, cs.n, max_branch_factor, status.ctypes.data_as(C.POINTER(C.c_int32)),
                                      exact.ctypes.data_as(u8p), threads)
    assert rc == 0
    return status, exact[:batch.n_regions]
