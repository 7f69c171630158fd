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


class MultiBatch:
    """A batch of MultiRegions as flat numpy arrays (avk_multi_batch): input i of region m owns variants
    [in_off[m*k + i], +in_cnt[m*k + i])."""

    FIELDS = ("region_id", "contig_idx", "start", "end", "in_off", "in_cnt", "var_pos", "var_type", "var_zyg", "var_raw_space",
              "a0_off", "a0_len", "a1_off", "a1_len", "allele_bytes")
    DTYPES = (np.uint64, np.uint32, np.uint64, np.uint64, np.uint64, np.uint32, np.uint64, np.uint8, np.uint8, np.uint32,
              np.uint64, np.uint32, np.uint64, np.uint32, np.uint8)

    def __init__(self, n_inputs, **arrays):
        self.n_inputs = int(n_inputs)
        for name, dt in zip(self.FIELDS, self.DTYPES):
            setattr(self, name, np.ascontiguousarray(arrays[name], dtype=dt))
        if self.allele_bytes.size == 0:
            self.allele_bytes = np.zeros(1, np.uint8)
        self.n_regions = int(self.region_id.size)
        self.n_variants = int(self.var_pos.size)
        assert self.in_off.size == self.n_regions * self.n_inputs and self.in_cnt.size == self.in_off.size

    @classmethod
    def from_regions(cls, multi_regions, n_inputs=None):
        """MultiRegion dicts {start, end, inputs:[variants...][, contig, region_id]}, the same number of inputs each"""
        n = len(multi_regions)
        k = len(multi_regions[0]["inputs"]) if n else (n_inputs or 2)
        assert all(len(mr["inputs"]) == k for mr in multi_regions), "regions of one batch must have the same number of inputs"
        # flatten through RegionBatch: region m contributes one pseudo-region per input (its variants as "truth")
        flat = RegionBatch.from_regions([{"start": mr["start"], "end": mr["end"], "contig": mr.get("contig", 0), "truth": inp, "query": []}
                                         for mr in multi_regions for inp in mr["inputs"]])
        return cls(k, region_id=[mr.get("region_id", m) for m, mr in enumerate(multi_regions)], contig_idx=[mr.get("contig", 0) for mr in multi_regions],
                   start=[mr["start"] for mr in multi_regions], end=[mr["end"] for mr in multi_regions], in_off=flat.t_off, in_cnt=flat.t_cnt,
                   var_pos=flat.var_pos, var_type=flat.var_type, var_zyg=flat.var_zyg, var_raw_space=flat.var_raw_space, a0_off=flat.a0_off,
                   a0_len=flat.a0_len, a1_off=flat.a1_off, a1_len=flat.a1_len, allele_bytes=flat.allele_bytes)

    def c_struct(self):
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        return AvkMultiBatch(self.n_regions, self.n_inputs, P(self.region_id, C.c_uint64), P(self.contig_idx, C.c_uint32), P(self.start, C.c_uint64),
                             P(self.end, C.c_uint64), P(self.in_off, C.c_uint64), P(self.in_cnt, C.c_uint32), self.n_variants, P(self.var_pos, C.c_uint64),
                             P(self.var_type, C.c_uint8), P(self.var_zyg, C.c_uint8), P(self.var_raw_space, C.c_uint32), P(self.a0_off, C.c_uint64),
                             P(self.a0_len, C.c_uint32), P(self.a1_off, C.c_uint64), P(self.a1_len, C.c_uint32), P(self.allele_bytes, C.c_uint8),
                             self.allele_bytes.size)

    def regions(self):
        """back to MultiRegion dicts (variants as (pos, ref, alt, type code, zygosity code, raw_space) tuples)"""
        ab = self.allele_bytes.tobytes()
        out = []
        k = self.n_inputs
        for m in range(self.n_regions):
            inputs = []
            for i in range(k):
                o, c = int(self.in_off[m * k + i]), int(self.in_cnt[m * k + i])
                inputs.append([(int(self.var_pos[v]), ab[int(self.a0_off[v]):int(self.a0_off[v]) + int(self.a0_len[v])],
                                ab[int(self.a1_off[v]):int(self.a1_off[v]) + int(self.a1_len[v])], int(self.var_type[v]), int(self.var_zyg[v]),
                                int(self.var_raw_space[v]))
                               for v in range(o, o + c)])
            out.append({"region_id": int(self.region_id[m]), "contig": int(self.contig_idx[m]), "start": int(self.start[m]), "end": int(self.end[m]),
                        "inputs": inputs})
        return out


class AvkPackedMultiBatch(C.Structure):
    _p = C.POINTER
    _fields_ = [("n_regions", C.c_uint64), ("n_inputs", C.c_uint32), ("contig_idx", _p(C.c_uint16)), ("start", _p(C.c_uint32)), ("len", _p(C.c_uint16)),
                ("in_cnt", _p(C.c_uint8)), ("n_variants", C.c_uint64), ("var_rel_pos", _p(C.c_uint16)), ("var_type_zyg", _p(C.c_uint8)),
                ("a0_len", _p(C.c_uint8)), ("a1_len", _p(C.c_uint8)), ("var_raw_space", _p(C.c_uint32)), ("allele_bytes", _p(C.c_uint8)),
                ("allele_bytes_len", C.c_uint64)]


class PackedMultiBatch:
    """The same batch in the library's packed form (avk_packed_multi_batch): 8 + k bytes per region, 5 per call plus the allele bytes; every offset is implied
    by order.  `from_multi` checks the constraints (calls back to back in region / input order, alleles back to back in call order, narrow fields wide enough)."""

    FIELDS = ("contig_idx", "start", "len", "in_cnt", "var_rel_pos", "var_type_zyg", "a0_len", "a1_len", "var_raw_space", "allele_bytes")
    DTYPES = (np.uint16, np.uint32, np.uint16, np.uint8, np.uint16, np.uint8, np.uint8, np.uint8, np.uint32, np.uint8)

    def __init__(self, n_inputs, **arrays):
        self.n_inputs = int(n_inputs)
        for name, dt in zip(self.FIELDS, self.DTYPES):
            a = arrays.get(name)
            setattr(self, name, None if a is None else np.ascontiguousarray(a, dtype=dt))
        self.n_regions = int(self.start.size)
        self.n_variants = int(self.var_rel_pos.size)
        assert self.in_cnt.size == self.n_regions * self.n_inputs

    @classmethod
    def from_multi(cls, mb, keep_raw_space=None):
        n, nv, k = mb.n_regions, mb.n_variants, mb.n_inputs
        cnt = mb.in_cnt.astype(np.int64)
        ioff = np.concatenate([[0], np.cumsum(cnt)])
        alen = mb.a0_len.astype(np.int64) + mb.a1_len
        aoff = np.concatenate([[0], np.cumsum(alen)])
        nbytes = int(aoff[-1])
        ok = (np.array_equal(mb.in_off, ioff[:-1]) and int(ioff[-1]) == nv and np.array_equal(mb.a0_off, aoff[:-1]) and np.array_equal(mb.a1_off, aoff[:-1] + mb.a0_len) and
              (nv == 0 or nbytes == mb.allele_bytes.size) and bool(np.all(mb.end >= mb.start)) and
              (n == 0 or (int((mb.end - mb.start).max()) < 65536 and int(mb.start.max()) < 2 ** 32 and int(cnt.max()) < 256 and int(mb.contig_idx.max()) < 65536)) and
              (nv == 0 or (int(mb.a0_len.max()) < 256 and int(mb.a1_len.max()) < 256 and int(mb.var_type.max()) < 16 and int(mb.var_zyg.max()) < 16)))
        rel = None
        if ok:
            per_region = cnt.reshape(n, k).sum(axis=1) if n else np.zeros(0, np.int64)
            rel = mb.var_pos.astype(np.int64) - np.repeat(mb.start.astype(np.int64), per_region)
            ok = nv == 0 or (int(rel.min()) >= 0 and int(rel.max()) < 65536)
        if not ok:
            raise ValueError("the batch does not satisfy the constraints of the packed form (include/aardvark_amd.h: avk_packed_multi_batch)")
        if keep_raw_space is None:
            keep_raw_space = not np.array_equal(mb.var_raw_space, np.maximum(mb.a0_len, mb.a1_len))
        return cls(k, contig_idx=mb.contig_idx, start=mb.start, len=mb.end - mb.start, in_cnt=mb.in_cnt, var_rel_pos=rel, var_type_zyg=mb.var_type | (mb.var_zyg << 4),
                   a0_len=mb.a0_len, a1_len=mb.a1_len, var_raw_space=mb.var_raw_space if keep_raw_space else None,
                   allele_bytes=mb.allele_bytes[:nbytes] if nv else np.zeros(1, np.uint8))

    def widen(self):
        """back to the wide form (what the library's dp_widen_packed_multi kernel writes on the device)"""
        n, k = self.n_regions, self.n_inputs
        cnt = self.in_cnt.astype(np.int64)
        ioff = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        aoff = np.concatenate([[0], np.cumsum(self.a0_len.astype(np.int64) + self.a1_len)])[:-1]
        per_region = cnt.reshape(n, k).sum(axis=1) if n else np.zeros(0, np.int64)
        raw = self.var_raw_space if self.var_raw_space is not None else np.maximum(self.a0_len, self.a1_len)
        return MultiBatch(k, region_id=np.arange(n), contig_idx=self.contig_idx if self.contig_idx is not None else np.zeros(n), start=self.start,
                          end=self.start.astype(np.uint64) + self.len, in_off=ioff, in_cnt=self.in_cnt, var_pos=np.repeat(self.start.astype(np.int64), per_region) + self.var_rel_pos,
                          var_type=self.var_type_zyg & 15, var_zyg=self.var_type_zyg >> 4, var_raw_space=raw, a0_off=aoff, a0_len=self.a0_len, a1_off=aoff + self.a0_len,
                          a1_len=self.a1_len, allele_bytes=self.allele_bytes)

    def nbytes(self):
        return sum(getattr(self, f).nbytes for f in self.FIELDS if getattr(self, f) is not None)

    def c_struct(self):
        b = AvkPackedMultiBatch()
        b.n_regions, b.n_inputs, b.n_variants = self.n_regions, self.n_inputs, self.n_variants
        for name, ct in (("contig_idx", C.c_uint16), ("start", C.c_uint32), ("len", C.c_uint16), ("in_cnt", C.c_uint8), ("var_rel_pos", C.c_uint16), ("var_type_zyg", C.c_uint8),
                         ("a0_len", C.c_uint8), ("a1_len", C.c_uint8), ("var_raw_space", C.c_uint32), ("allele_bytes", C.c_uint8)):
            a = getattr(self, name)
            if a is not None:
                setattr(b, name, a.ctypes.data_as(C.POINTER(ct)))
        b.allele_bytes_len = int(self.allele_bytes.size) if self.n_variants else 0
        return b


class MergeResult:
    """status / classification (AVK_MERGE_*) / members per region, the outputs of avk_merge_batch"""

    def __init__(self, status, classification, members, n_inputs):
        self.status, self.classification, self.members, self.n_inputs = status, classification, members, n_inputs

    def decoded(self):
        """[(status, None) | (0, ("identical",) / ("no_conflict", indices) / ("majority", indices) / ("conflict_select", index) / ("different",))]"""
        out = []
        for m in range(len(self.status)):
            if self.status[m] != 0:
                out.append((int(self.status[m]), None))
                continue
            name = CLASSES[int(self.classification[m])]
            if name in ("no_conflict", "majority"):
                out.append((0, (name, [i for i in range(self.n_inputs) if (int(self.members[m]) >> i) & 1])))
            elif name == "conflict_select":
                out.append((0, (name, int(self.members[m]))))
            else:
                out.append((0, (name,)))
        return out


def pinned_multi_batch(ctx, mb):
    """a copy of a MultiBatch / PackedMultiBatch whose arrays live in pinned memory (avk_host_alloc): avk_merge_batch / avk_merge_packed then copy them by DMA
    instead of through the bounce buffer"""
    arrays = {}
    for name in type(mb).FIELDS:
        a = getattr(mb, name)
        if a is None:
            continue
        out = ctx.host_array(a.shape, a.dtype)
        out[...] = a
        arrays[name] = out
    return type(mb)(mb.n_inputs, **arrays)


def merge_multi_batch(ctx, mb, config=None):
    """avk_merge_batch on a MultiBatch, avk_merge_packed on a PackedMultiBatch: all of solve_merge_region on the GPU -> MergeResult"""
    config = config or MergeConfig()
    n = mb.n_regions
    cfg = AvkMergeConfig(config.max_branch_factor, int(config.no_conflict_enabled), int(config.majority_voting_enabled),
                         -1 if config.conflict_selection is None else int(config.conflict_selection))
    st = np.zeros(max(n, 1), np.int32)
    cls = np.zeros(max(n, 1), np.uint8)
    members = np.zeros(max(n, 1), np.uint64)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    cb = mb.c_struct()
    packed = isinstance(mb, PackedMultiBatch)
    entry = ctx.lib.avk_merge_packed if packed else ctx.lib.avk_merge_batch
    entry.argtypes = [C.c_void_p, C.POINTER(AvkPackedMultiBatch if packed else AvkMultiBatch), C.POINTER(AvkMergeConfig), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    ctx._check(entry(ctx.handle, C.byref(cb), C.byref(cfg), P(st, C.c_int32), P(cls, C.c_uint8), P(members, C.c_uint64)))
    return MergeResult(st[:n], cls[:n], members[:n], mb.n_inputs)


def merge_batch(ctx, multi_regions, config=None):
    """avk_merge_batch: all of solve_merge_region for a list of MultiRegion dicts (same number of inputs each) in one library
    call — pairs on the GPU, classification on the host.  Same return shape as solve_merge_regions."""
    if not multi_regions:
        return []
    return merge_multi_batch(ctx, MultiBatch.from_regions(multi_regions), config).decoded()


# ---- merge on several GPUs (include/aardvark_amd.h: avk_packed_multi_shard_*, avk_merge_counts*; reference src/main.rs:463-478, src/writers/merge_summary.rs:12-18) ----

MERGE_COUNTS_MAX_INPUTS = 10


def _shard_api(lib):
    if getattr(lib, "_avk_multi_shard_api", False):
        return lib
    P = C.POINTER
    lib.avk_packed_multi_shard_make.argtypes = [P(AvkPackedMultiBatch), P(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32, P(C.c_void_p)]
    lib.avk_packed_multi_shard_batch.restype = P(AvkPackedMultiBatch)
    lib.avk_packed_multi_shard_batch.argtypes = [C.c_void_p]
    lib.avk_packed_multi_shard_regions.restype = C.c_uint64
    lib.avk_packed_multi_shard_regions.argtypes = [C.c_void_p, P(P(C.c_uint64))]
    lib.avk_packed_multi_shard_scatter.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_uint8), P(C.c_uint64), P(C.c_int32), P(C.c_uint8), P(C.c_uint64)]
    lib.avk_packed_multi_shard_free.argtypes = [C.c_void_p]
    lib.avk_merge_counts_len.restype = C.c_uint64
    lib.avk_merge_counts_len.argtypes = [C.c_uint32]
    lib.avk_merge_counts_reason.restype = C.c_uint32
    lib.avk_merge_counts_reason.argtypes = [C.c_uint32, C.c_uint8, C.c_uint64]
    lib.avk_merge_counts.argtypes = [P(AvkPackedMultiBatch), P(C.c_int32), P(C.c_uint8), P(C.c_uint64), P(C.c_uint64)]
    lib.avk_counts_allreduce.argtypes = [C.c_void_p, C.c_void_p, P(C.c_uint64), C.c_uint64]
    lib._avk_multi_shard_api = True
    return lib


def shard_packed_multi(lib, pmb, region_id, rank, world):
    """the regions of the packed multi-region batch that rank `rank` of `world` owns (avk_packed_multi_shard_make: shard = avk_region_shard(region_id), the rule of
    dist.region_hash) -> (PackedMultiBatch with copies of the shard's arrays, the regions' indices in the whole batch)"""
    lib = _shard_api(lib)
    ids = np.ascontiguousarray(region_id, np.uint64)
    cb = pmb.c_struct()
    h = C.c_void_p()
    rc = lib.avk_packed_multi_shard_make(C.byref(cb), ids.ctypes.data_as(C.POINTER(C.c_uint64)), 0, rank, world, C.byref(h))
    if rc:
        raise ValueError("avk_packed_multi_shard_make failed (%d)" % rc)
    try:
        b = lib.avk_packed_multi_shard_batch(h).contents
        n, nv, na, k = int(b.n_regions), int(b.n_variants), int(b.allele_bytes_len), int(b.n_inputs)
        take = lambda ptr, m, dt: np.ctypeslib.as_array(ptr, shape=(max(m, 1),))[:m].astype(dt).copy() if ptr else None
        shard = PackedMultiBatch(k, contig_idx=take(b.contig_idx, n, np.uint16), start=take(b.start, n, np.uint32), len=take(b.len, n, np.uint16),
                                 in_cnt=take(b.in_cnt, n * k, np.uint8), var_rel_pos=take(b.var_rel_pos, nv, np.uint16), var_type_zyg=take(b.var_type_zyg, nv, np.uint8),
                                 a0_len=take(b.a0_len, nv, np.uint8), a1_len=take(b.a1_len, nv, np.uint8), var_raw_space=take(b.var_raw_space, nv, np.uint32),
                                 allele_bytes=take(b.allele_bytes, na, np.uint8) if nv else np.zeros(1, np.uint8))
        idx = C.POINTER(C.c_uint64)()
        m = int(lib.avk_packed_multi_shard_regions(h, C.byref(idx)))
        index = np.ctypeslib.as_array(idx, shape=(max(m, 1),))[:m].copy()
    finally:
        lib.avk_packed_multi_shard_free(h)
    return shard, index


def merge_counts_len(lib, n_inputs):
    return int(_shard_api(lib).avk_merge_counts_len(n_inputs))


def merge_counts(lib, pmb, result, counts=None):
    """MergeSummaryWriter::add_merge_benchmark over a solved packed batch as a dense block of sums (avk_merge_counts), ADDED to `counts`"""
    lib = _shard_api(lib)
    n = merge_counts_len(lib, pmb.n_inputs)
    if n == 0:
        raise ValueError("dense summary counters exist for 2..%d inputs" % MERGE_COUNTS_MAX_INPUTS)
    if counts is None:
        counts = np.zeros(n, np.uint64)
    assert counts.dtype == np.uint64 and counts.size == n and counts.flags.c_contiguous
    P = lambda a, t: np.ascontiguousarray(a).ctypes.data_as(C.POINTER(t))
    st, cls, mem = (np.ascontiguousarray(result.status, np.int32), np.ascontiguousarray(result.classification, np.uint8), np.ascontiguousarray(result.members, np.uint64))
    if pmb.n_regions == 0:
        return counts
    cb = pmb.c_struct()
    rc = lib.avk_merge_counts(C.byref(cb), P(st, C.c_int32), P(cls, C.c_uint8), P(mem, C.c_uint64), counts.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc:
        raise ValueError("avk_merge_counts failed (%d)" % rc)
    return counts
