"""Loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

from aardvark_amd._abi import (AvkCompareConfig, AvkRegionBatch, AvkResultBatch, RegionBatch, ResultBatch)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


def build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle.cpp", "oracle.h")] + [os.path.join(ROOT, "include", "aardvark_amd.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        import fcntl
        with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:  # (pytest-xdist workers: one builds, the others wait)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def b2p(b):
    """bytes -> (uint8*, len); keeps a reference alive on the returned array."""
    arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))
    return arr, len(b)


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build())
    lib.orc_wfa_ed.restype = C.c_uint64
    lib.orc_wfa_ed.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
    lib.orc_edit_distance.restype = C.c_uint64
    lib.orc_edit_distance.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
    lib.orc_dwfa_new.restype = C.c_void_p
    lib.orc_dwfa_new.argtypes = [C.c_uint64]
    lib.orc_dwfa_free.argtypes = [C.c_void_p]
    lib.orc_dwfa_clone.restype = C.c_void_p
    lib.orc_dwfa_clone.argtypes = [C.c_void_p]
    lib.orc_dwfa_update.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p, C.c_uint64]
    lib.orc_dwfa_finalize.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p, C.c_uint64]
    lib.orc_dwfa_ed.restype = C.c_uint64
    lib.orc_dwfa_ed.argtypes = [C.c_void_p]
    lib.orc_dwfa_wavefront.restype = C.c_uint64
    lib.orc_dwfa_wavefront.argtypes = [C.c_void_p, u64p, C.c_uint64]
    lib.orc_dwfa_equal.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_hapnode_new.restype = C.c_void_p
    lib.orc_hapnode_new.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
    lib.orc_hapnode_free.argtypes = [C.c_void_p]
    lib.orc_hapnode_extend.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64,
                                       C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int)]
    lib.orc_hapnode_finalize.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint64]
    for f in ("orc_hapnode_ed", "orc_hapnode_skip"):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = [C.c_void_p, C.c_int]
    lib.orc_hapnode_cost.restype = C.c_uint64
    lib.orc_hapnode_cost.argtypes = [C.c_void_p]
    for f in ("orc_hapnode_seq", "orc_hapnode_alleles"):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = [C.c_void_p, C.c_int, C.c_int, u8p, C.c_uint64]
    lib.orc_optimize_sequences.restype = C.c_int64
    lib.orc_optimize_sequences.argtypes = [C.POINTER(AvkRegionBatch), C.c_uint64, u8p, C.c_uint64, C.c_uint32, C.c_uint32,
                                           u64p, u64p, u8p, u8p]
    lib.orc_last_sequence.restype = C.c_uint64
    lib.orc_last_sequence.argtypes = [C.c_uint32, C.c_int, u8p, C.c_uint64]
    lib.orc_optimize_gt_alleles.restype = C.c_int64
    lib.orc_optimize_gt_alleles.argtypes = [C.POINTER(AvkRegionBatch), C.c_uint64, u8p, C.c_uint64, u8p, u8p, u8p, u8p]
    lib.orc_compare_batch.argtypes = [C.POINTER(AvkRegionBatch), C.POINTER(u8p), u64p, C.c_uint32,
                                      C.POINTER(AvkCompareConfig), C.POINTER(AvkResultBatch), C.c_int]
    lib.orc_basepair_compare.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
    lib.orc_optimize_pairs_batch.argtypes = [C.POINTER(AvkRegionBatch), C.POINTER(u8p), u64p, C.c_uint32, C.c_uint32,
                                             C.POINTER(C.c_int32), u8p, C.c_int]
    lib.orc_last_stats.argtypes = [u64p]
    lib.orc_bench.argtypes = [C.POINTER(AvkRegionBatch), C.POINTER(u8p), u64p, C.c_uint32, C.POINTER(AvkCompareConfig), C.c_int, C.c_int,
                              C.POINTER(C.c_double), u64p]
    lib.orc_group_add_truth.argtypes = [u64p, C.c_uint64, C.c_uint8, C.c_uint8]
    lib.orc_group_add_query.argtypes = [u64p, C.c_uint64, C.c_uint8, C.c_uint8]
    lib.orc_group_swap.argtypes = [u64p, u64p]
    lib.orc_variant_metrics.argtypes = [C.c_uint8, C.c_uint8, u8p, u8p]
    lib.orc_generate_haplotype_sequence.restype = C.c_int64
    lib.orc_generate_haplotype_sequence.argtypes = [C.POINTER(AvkRegionBatch), C.c_uint64, C.c_int, u8p, C.c_uint64, u8p,
                                                    C.c_int, u8p, C.c_uint64, u64p]
    _lib = lib
    return lib


class ContigSet:
    """Keeps contig byte arrays alive and exposes the (uint8**, uint64*) pair the oracle takes."""

    def __init__(self, contigs):
        self.arrs = [np.frombuffer(c if isinstance(c, (bytes, bytearray)) else bytes(c), dtype=np.uint8) if not isinstance(c, np.ndarray)
                     else np.ascontiguousarray(c, dtype=np.uint8) for c in contigs]
        self.ptrs = (u8p * len(self.arrs))(*[a.ctypes.data_as(u8p) if a.size else C.cast(None, u8p) for a in self.arrs])
        self.lens = (C.c_uint64 * len(self.arrs))(*[a.size for a in self.arrs])
        self.n = len(self.arrs)


def compare_batch(lib, batch, contigs, max_branch_factor=50, sequences=False, exact_shortcut=False, threads=1, group_metrics=True):
    """Runs the oracle's solve_compare_region over a RegionBatch; returns a ResultBatch (group_metrics=False: no per-region 13x22
    blocks, 1144 bytes per region — the whole-genome batches of bench.py)."""
    cs = contigs if isinstance(contigs, ContigSet) else ContigSet(contigs)
    res = ResultBatch(batch, sequences=sequences, group_metrics=group_metrics)
    cfg = AvkCompareConfig(max_branch_factor, 1 if sequences else 0, 1 if exact_shortcut else 0)
    cb, ro = batch.c_struct(), res.c_struct()
    rc = lib.orc_compare_batch(C.byref(cb), cs.ptrs, cs.lens, cs.n, C.byref(cfg), C.byref(ro), threads)
    assert rc == 0
    return res


def stats(lib):
    out = (C.c_uint64 * 16)()
    lib.orc_last_stats(out)
    names = ["max_pops_a", "max_queue_a", "max_pops_b", "max_queue_b", "max_ed", "max_optima", "total_pops_a", "total_pops_b", "total_wfa", "incr_mismatch", "byte_compares"]
    return {n: int(out[i]) for i, n in enumerate(names)}


def bench(lib, batch, contigs, threads, reps, max_branch_factor=50):
    """Timed oracle passes (CPU baseline): returns (seconds, regions_per_second)."""
    cs = contigs if isinstance(contigs, ContigSet) else ContigSet(contigs)
    cfg = AvkCompareConfig(max_branch_factor, 0, 0)
    cb = batch.c_struct()
    sec, chk = C.c_double(0), C.c_uint64(0)
    rc = lib.orc_bench(C.byref(cb), cs.ptrs, cs.lens, cs.n, C.byref(cfg), threads, reps, C.byref(sec), C.byref(chk))
    assert rc == 0
    return sec.value, batch.n_regions * reps / sec.value


def optimize_pairs(lib, batch, contigs, max_branch_factor=50, threads=1):
    cs = contigs if isinstance(contigs, ContigSet) else ContigSet(contigs)
    status = np.full(batch.n_regions, -1, np.int32)
    exact = np.zeros(max(batch.n_regions, 1), np.uint8)
    cb = batch.c_struct()
    rc = lib.orc_optimize_pairs_batch(C.byref(cb), cs.ptrs, cs.lens, cs.n, max_branch_factor, status.ctypes.data_as(C.POINTER(C.c_int32)),
                                      exact.ctypes.data_as(u8p), threads)
    assert rc == 0
    return status, exact[:batch.n_regions]
