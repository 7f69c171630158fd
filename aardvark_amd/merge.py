"""Host-side mirror of `solve_merge_region` (reference src/merge_solver.rs:110-200).

The device does the expensive part — the all-pairs `optimize_sequences` exact-match test
(avk_optimize_pairs_batch); the classification on top of the pair matrix is the library's host function
avk_merge_classify.  avk_merge_batch does both in one call for a flat avk_multi_batch.
"""
import ctypes as C

import numpy as np

from ._abi import ZYG, RegionBatch


class MergeConfig:
    """MergeConfig (src/merge_solver.rs:62-84), same defaults"""

    def __init__(self, no_conflict_enabled=False, majority_voting_enabled=False, conflict_selection=None, max_branch_factor=50):
        self.no_conflict_enabled = no_conflict_enabled
        self.majority_voting_enabled = majority_voting_enabled
        self.conflict_selection = conflict_selection
        self.max_branch_factor = max_branch_factor


def pair_batch(multi_regions):
    """One CompareRegion-shaped item per (i < j) input pair of every MultiRegion
    (src/data_types/multi_region.rs): dicts {start, end, inputs:[variants...][, contig, region_id]}."""
    regions, owner = [], []
    for m, mr in enumerate(multi_regions):
        k = len(mr["inputs"])
        for i in range(k):
            for j in range(i + 1, k):
                regions.append({"start": mr["start"], "end": mr["end"], "contig": mr.get("contig", 0),
                                "truth": mr["inputs"][i], "query": mr["inputs"][j]})
                owner.append((m, i, j))
    return RegionBatch.from_regions(regions), owner


CLASSES = {0: "different", 1: "identical", 2: "no_conflict", 3: "majority", 4: "conflict_select"}


class AvkMergeConfig(C.Structure):
    _fields_ = [("max_branch_factor", C.c_uint32), ("no_conflict_enabled", C.c_uint32), ("majority_voting_enabled", C.c_uint32),
                ("conflict_selection", C.c_int32)]


def solve_merge_regions(pairs_fn, multi_regions, config=None):
    """pairs_fn(batch, max_branch_factor) -> (status[], is_exact_match[]) is the device call
    (Context.optimize_pairs); the classification on top of the pair matrix is the library's host function
    avk_merge_classify (C++, aardvark_amd/csrc/avk_host.hip).  All regions must have the same number of inputs.
    Returns one (status, classification) per region where classification is
    ("identical",) / ("no_conflict", indices) / ("majority", indices) / ("conflict_select", index) /
    ("different",), the reference's MergeClassification (src/data_types/merge_benchmark.rs:5-14)."""
    from .api import load_library
    config = config or MergeConfig()
    n = len(multi_regions)
    if n == 0:
        return []
    k = len(multi_regions[0]["inputs"])
    assert all(len(mr["inputs"]) == k for mr in multi_regions), "regions of one call must have the same number of inputs"
    batch, owner = pair_batch(multi_regions)
    status, exact = pairs_fn(batch, config.max_branch_factor) if batch.n_regions else ([], [])
    in_cnt = np.array([[len(v) for v in mr["inputs"]] for mr in multi_regions], np.uint32).reshape(-1)
    unknown = np.array([any((v[4] if len(v) > 4 else "Unknown") in ("Unknown", ZYG["Unknown"]) for vs in mr["inputs"] for v in vs)
                        for mr in multi_regions], np.uint8)
    pst = np.ascontiguousarray(status, np.int32) if len(status) else np.zeros(1, np.int32)
    pex = np.ascontiguousarray(exact, np.uint8) if len(exact) else np.zeros(1, np.uint8)
    cfg = AvkMergeConfig(config.max_branch_factor, int(config.no_conflict_enabled), int(config.majority_voting_enabled),
                         -1 if config.conflict_selection is None else int(config.conflict_selection))
    st = np.zeros(n, np.int32)
    cls = np.zeros(n, np.uint8)
    members = np.zeros(n, np.uint64)
    lib = load_library()
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib.avk_merge_classify(C.c_uint64(n), C.c_uint32(k), P(in_cnt, C.c_uint32), P(unknown, C.c_uint8), P(pst, C.c_int32), P(pex, C.c_uint8),
                                C.byref(cfg), P(st, C.c_int32), P(cls, C.c_uint8), P(members, C.c_uint64))
    assert rc == 0
    out = []
    for m in range(n):
        if st[m] != 0:
            out.append((int(st[m]), None))
            continue
        name = CLASSES[int(cls[m])]
        if name in ("no_conflict", "majority"):
            out.append((0, (name, [i for i in range(k) if (int(members[m]) >> i) & 1])))
        elif name == "conflict_select":
            out.append((0, (name, int(members[m]))))
        else:
            out.append((0, (name,)))
    return out


class AvkMultiBatch(C.Structure):
    _p = C.POINTER
    _fields_ = [("n_regions", C.c_uint64), ("n_inputs", C.c_uint32), ("region_id", _p(C.c_uint64)), ("contig_idx", _p(C.c_uint32)),
                ("start", _p(C.c_uint64)), ("end", _p(C.c_uint64)), ("in_off", _p(C.c_uint64)), ("in_cnt", _p(C.c_uint32)),
                ("n_variants", C.c_uint64), ("var_pos", _p(C.c_uint64)), ("var_type", _p(C.c_uint8)), ("var_zyg", _p(C.c_uint8)),
                ("var_raw_space", _p(C.c_uint32)), ("a0_off", _p(C.c_uint64)), ("a0_len", _p(C.c_uint32)), ("a1_off", _p(C.c_uint64)),
                ("a1_len", _p(C.c_uint32)), ("allele_bytes", _p(C.c_uint8)), ("allele_bytes_len", C.c_uint64)]


def merge_batch(ctx, multi_regions, config=None):
    """avk_merge_batch: all of solve_merge_region for a list of MultiRegion dicts (same number of inputs each) in one library
    call — pairs on the GPU, classification on the host.  Same return shape as solve_merge_regions."""
    config = config or MergeConfig()
    n = len(multi_regions)
    k = len(multi_regions[0]["inputs"]) if n else 2
    # flatten through RegionBatch: region m contributes one pseudo-region per input (its variants as "truth")
    flat = RegionBatch.from_regions([{"start": mr["start"], "end": mr["end"], "contig": mr.get("contig", 0), "truth": inp, "query": []}
                                     for mr in multi_regions for inp in mr["inputs"]])
    g = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    start = g([mr["start"] for mr in multi_regions], np.uint64)
    end = g([mr["end"] for mr in multi_regions], np.uint64)
    rid = g(range(n), np.uint64)
    cidx = g([mr.get("contig", 0) for mr in multi_regions], np.uint32)
    in_off, in_cnt = g(flat.t_off, np.uint64), g(flat.t_cnt, np.uint32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    mb = AvkMultiBatch(n, k, P(rid, C.c_uint64), P(cidx, C.c_uint32), P(start, C.c_uint64), P(end, C.c_uint64), P(in_off, C.c_uint64), P(in_cnt, C.c_uint32),
                       flat.n_variants, P(flat.var_pos, C.c_uint64), P(flat.var_type, C.c_uint8), P(flat.var_zyg, C.c_uint8), P(flat.var_raw_space, C.c_uint32),
                       P(flat.a0_off, C.c_uint64), P(flat.a0_len, C.c_uint32), P(flat.a1_off, C.c_uint64), P(flat.a1_len, C.c_uint32),
                       P(flat.allele_bytes, C.c_uint8), flat.allele_bytes.size)
    cfg = AvkMergeConfig(config.max_branch_factor, int(config.no_conflict_enabled), int(config.majority_voting_enabled),
                         -1 if config.conflict_selection is None else int(config.conflict_selection))
    st = np.zeros(max(n, 1), np.int32)
    cls = np.zeros(max(n, 1), np.uint8)
    members = np.zeros(max(n, 1), np.uint64)
    ctx.lib.avk_merge_batch.argtypes = [C.c_void_p, C.POINTER(AvkMultiBatch), C.POINTER(AvkMergeConfig), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    ctx._check(ctx.lib.avk_merge_batch(ctx.handle, C.byref(mb), C.byref(cfg), P(st, C.c_int32), P(cls, C.c_uint8), P(members, C.c_uint64)))
    out = []
    for m in range(n):
        if st[m] != 0:
            out.append((int(st[m]), None))
            continue
        name = CLASSES[int(cls[m])]
        if name in ("no_conflict", "majority"):
            out.append((0, (name, [i for i in range(k) if (int(members[m]) >> i) & 1])))
        elif name == "conflict_select":
            out.append((0, (name, int(members[m]))))
        else:
            out.append((0, (name,)))
    return out
