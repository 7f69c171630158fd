"""ctypes mirror of include/aardvark_amd.h plus the flat batch containers.

Plumbing only: the structures here are bit-for-bit the C-ABI structs; `RegionBatch` packs
Python-side regions into the structure-of-arrays layout the library takes and `ResultBatch`
owns the caller-allocated output arrays.
"""
import ctypes as C

import numpy as np

# ---- enums (ordinals of the reference's Rust enums; include/aardvark_amd.h) -------------
VARIANT_TYPES = [
    "Snv", "Insertion", "Deletion", "Indel", "SvInsertion", "SvDeletion", "SvDuplication",
    "SvInversion", "SvBreakend", "TrContraction", "TrExpansion", "Unknown",
]
VT = {n: i for i, n in enumerate(VARIANT_TYPES)}
ZYGOSITIES = [
    "Unknown", "HomozygousReference", "UnphasedHeterozygous", "PhasedHet01", "PhasedHet10",
    "HomozygousAlternate",
]
ZYG = {n: i for i, n in enumerate(ZYGOSITIES)}
CLASSES = ["UNK", "TP", "FN", "FP"]
CLS = {n: i for i, n in enumerate(CLASSES)}
ALLELE = {"UNK": 0, "REF": 1, "ALT": 2}

N_GROUPS = 13
N_FIELDS = 22
TALLY_LEN = N_GROUPS * N_FIELDS + 2
FIELDS = [
    "GT_TRUTH_TP", "GT_TRUTH_FN", "GT_QUERY_TP", "GT_QUERY_FP", "GT_TRUTH_FN_GT", "GT_QUERY_FP_GT",
    "HAP_TRUTH_TP", "HAP_TRUTH_FN", "HAP_QUERY_TP", "HAP_QUERY_FP",
    "WHAP_TRUTH_TP", "WHAP_TRUTH_FN", "WHAP_QUERY_TP", "WHAP_QUERY_FP",
    "BP_TRUTH_TP", "BP_TRUTH_FN", "BP_QUERY_TP", "BP_QUERY_FP",
    "RBP_TRUTH_TP", "RBP_TRUTH_FN", "RBP_QUERY_TP", "RBP_QUERY_FP",
]
F = {n: i for i, n in enumerate(FIELDS)}

ST_OK = 0
ST_NAMES = {
    0: "OK", 2: "BRANCH_FACTOR", 3: "NO_RESULTS", 4: "NO_GT_RESULT", 5: "UNKNOWN_ALLELE",
    6: "BAD_ZYGOSITY", 7: "VARIANT_METRICS", 8: "TRUTH_FP", 9: "RECORD_BP", 10: "SEQ_MISMATCH",
    11: "AUTOFAIL_OOB", 20: "INVALID_INPUT", 21: "CAPACITY",
}

_p = C.POINTER


class AvkRegionBatch(C.Structure):
    _fields_ = [
        ("n_regions", C.c_uint64),
        ("region_id", _p(C.c_uint64)),
        ("contig_idx", _p(C.c_uint32)),
        ("start", _p(C.c_uint64)),
        ("end", _p(C.c_uint64)),
        ("t_off", _p(C.c_uint64)),
        ("t_cnt", _p(C.c_uint32)),
        ("q_off", _p(C.c_uint64)),
        ("q_cnt", _p(C.c_uint32)),
        ("n_variants", C.c_uint64),
        ("var_pos", _p(C.c_uint64)),
        ("var_type", _p(C.c_uint8)),
        ("var_zyg", _p(C.c_uint8)),
        ("var_raw_space", _p(C.c_uint32)),
        ("a0_off", _p(C.c_uint64)),
        ("a0_len", _p(C.c_uint32)),
        ("a1_off", _p(C.c_uint64)),
        ("a1_len", _p(C.c_uint32)),
        ("allele_bytes", _p(C.c_uint8)),
        ("allele_bytes_len", C.c_uint64),
    ]


class AvkCompactBatch(C.Structure):
    _fields_ = [
        ("n_regions", C.c_uint64),
        ("contig_idx", _p(C.c_uint32)),
        ("start", _p(C.c_uint32)),
        ("len", _p(C.c_uint32)),
        ("v_off", _p(C.c_uint32)),
        ("t_cnt", _p(C.c_uint16)),
        ("q_cnt", _p(C.c_uint16)),
        ("n_variants", C.c_uint64),
        ("var_pos", _p(C.c_uint32)),
        ("var_type_zyg", _p(C.c_uint8)),
        ("a_off", _p(C.c_uint32)),
        ("a0_len", _p(C.c_uint32)),
        ("a1_len", _p(C.c_uint32)),
        ("var_raw_space", _p(C.c_uint32)),
        ("allele_bytes", _p(C.c_uint8)),
        ("allele_bytes_len", C.c_uint64),
    ]


class AvkPackedBatch(C.Structure):
    _fields_ = [
        ("n_regions", C.c_uint64),
        ("contig_idx", _p(C.c_uint16)),
        ("start", _p(C.c_uint32)),
        ("len", _p(C.c_uint16)),
        ("t_cnt", _p(C.c_uint8)),
        ("q_cnt", _p(C.c_uint8)),
        ("n_variants", C.c_uint64),
        ("var_rel_pos", _p(C.c_uint16)),
        ("var_type_zyg", _p(C.c_uint8)),
        ("a0_len", _p(C.c_uint8)),
        ("a1_len", _p(C.c_uint8)),
        ("var_raw_space", _p(C.c_uint32)),
        ("allele_bytes", _p(C.c_uint8)),
        ("allele_bytes_len", C.c_uint64),
    ]


class AvkCompareConfig(C.Structure):
    _fields_ = [
        ("max_branch_factor", C.c_uint32),
        ("enable_sequences", C.c_uint32),
        ("enable_exact_shortcut", C.c_uint32),
    ]


class AvkResultBatch(C.Structure):
    _fields_ = [
        ("status", _p(C.c_int32)),
        ("ed_h1", _p(C.c_uint32)),
        ("ed_h2", _p(C.c_uint32)),
        ("n_optima", _p(C.c_uint32)),
        ("type_present", _p(C.c_uint16)),
        ("group_metrics", _p(C.c_uint32)),
        ("var_expected", _p(C.c_uint8)),
        ("var_observed", _p(C.c_uint8)),
        ("var_class", _p(C.c_uint8)),
        ("var_zyg", _p(C.c_uint8)),
        ("seq_bytes", _p(C.c_uint8)),
        ("seq_off", _p(C.c_uint64)),
        ("seq_stride", _p(C.c_uint32)),
        ("seq_len", _p(C.c_uint32)),
        ("tally", _p(C.c_uint64)),
        ("bp_off", _p(C.c_uint32)),
        ("bp_groups", _p(C.c_uint32)),
        ("region_packed", _p(C.c_uint64)),
        ("var_packed", _p(C.c_uint8)),
        ("bp_packed", _p(C.c_uint32)),
        ("bp_spilled", _p(C.c_uint32)),
    ]


def _ptr(arr, ctype):
    return arr.ctypes.data_as(_p(ctype))


def seq_stride(start, end, a0_len, a1_len):
    """Upper bound of any haplotype length of a region (mirrors avk_seq_stride())."""
    grow = int(np.maximum(a1_len.astype(np.int64) - a0_len.astype(np.int64), 0).sum()) if len(a0_len) else 0
    return int(end - start) + grow


class RegionBatch:
    """Flat structure-of-arrays batch of CompareRegions (reference
    src/data_types/compare_region.rs:13-26).  Build with `from_regions` (a list of dicts) or
    directly from numpy arrays via the constructor."""

    def __init__(self, region_id, contig_idx, start, end, t_off, t_cnt, q_off, q_cnt,
                 var_pos, var_type, var_zyg, var_raw_space, a0_off, a0_len, a1_off, a1_len, allele_bytes):
        g = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.region_id = g(region_id, np.uint64)
        self.contig_idx = g(contig_idx, np.uint32)
        self.start = g(start, np.uint64)
        self.end = g(end, np.uint64)
        self.t_off = g(t_off, np.uint64)
        self.t_cnt = g(t_cnt, np.uint32)
        self.q_off = g(q_off, np.uint64)
        self.q_cnt = g(q_cnt, np.uint32)
        self.var_pos = g(var_pos, np.uint64)
        self.var_type = g(var_type, np.uint8)
        self.var_zyg = g(var_zyg, np.uint8)
        self.var_raw_space = g(var_raw_space, np.uint32)
        self.a0_off = g(a0_off, np.uint64)
        self.a0_len = g(a0_len, np.uint32)
        self.a1_off = g(a1_off, np.uint64)
        self.a1_len = g(a1_len, np.uint32)
        self.allele_bytes = g(allele_bytes, np.uint8)
        if self.allele_bytes.size == 0:
            self.allele_bytes = np.zeros(1, np.uint8)
        self.n_regions = int(self.start.size)
        self.n_variants = int(self.var_pos.size)

    @classmethod
    def from_regions(cls, regions):
        """regions: list of dicts {start, end, truth, query[, contig, region_id]}; a variant is
        (pos, allele0, allele1, type, zygosity[, raw_allele_space]) with type/zygosity by name
        or ordinal and alleles as str/bytes."""
        rid, cidx, st, en, t_off, t_cnt, q_off, q_cnt = [], [], [], [], [], [], [], []
        vpos, vtype, vzyg, vraw, a0o, a0l, a1o, a1l = [], [], [], [], [], [], [], []
        blob = bytearray()

        def add(v):
            pos, a0, a1, ty = v[0], v[1], v[2], v[3]
            zy = v[4] if len(v) > 4 else "Unknown"
            a0 = a0.encode() if isinstance(a0, str) else bytes(a0)
            a1 = a1.encode() if isinstance(a1, str) else bytes(a1)
            vpos.append(pos)
            vtype.append(VT[ty] if isinstance(ty, str) else int(ty))
            vzyg.append(ZYG[zy] if isinstance(zy, str) else int(zy))
            vraw.append(v[5] if len(v) > 5 else max(len(a0), len(a1)))
            a0o.append(len(blob)); a0l.append(len(a0)); blob.extend(a0)
            a1o.append(len(blob)); a1l.append(len(a1)); blob.extend(a1)

        for i, r in enumerate(regions):
            rid.append(r.get("region_id", i))
            cidx.append(r.get("contig", 0))
            st.append(r["start"]); en.append(r["end"])
            t_off.append(len(vpos)); t_cnt.append(len(r["truth"]))
            for v in r["truth"]:
                add(v)
            q_off.append(len(vpos)); q_cnt.append(len(r["query"]))
            for v in r["query"]:
                add(v)
        return cls(rid, cidx, st, en, t_off, t_cnt, q_off, q_cnt, vpos, vtype, vzyg, vraw,
                   a0o, a0l, a1o, a1l, np.frombuffer(bytes(blob), dtype=np.uint8))

    def c_struct(self):
        b = AvkRegionBatch()
        b.n_regions = self.n_regions
        b.region_id = _ptr(self.region_id, C.c_uint64)
        b.contig_idx = _ptr(self.contig_idx, C.c_uint32)
        b.start = _ptr(self.start, C.c_uint64)
        b.end = _ptr(self.end, C.c_uint64)
        b.t_off = _ptr(self.t_off, C.c_uint64)
        b.t_cnt = _ptr(self.t_cnt, C.c_uint32)
        b.q_off = _ptr(self.q_off, C.c_uint64)
        b.q_cnt = _ptr(self.q_cnt, C.c_uint32)
        b.n_variants = self.n_variants
        b.var_pos = _ptr(self.var_pos, C.c_uint64)
        b.var_type = _ptr(self.var_type, C.c_uint8)
        b.var_zyg = _ptr(self.var_zyg, C.c_uint8)
        b.var_raw_space = _ptr(self.var_raw_space, C.c_uint32)
        b.a0_off = _ptr(self.a0_off, C.c_uint64)
        b.a0_len = _ptr(self.a0_len, C.c_uint32)
        b.a1_off = _ptr(self.a1_off, C.c_uint64)
        b.a1_len = _ptr(self.a1_len, C.c_uint32)
        b.allele_bytes = _ptr(self.allele_bytes, C.c_uint8)
        b.allele_bytes_len = int(self.allele_bytes.size)
        return b

    def seq_strides(self):
        out = np.zeros(self.n_regions, np.uint32)
        for r in range(self.n_regions):
            lo = int(self.t_off[r]); hi = lo + int(self.t_cnt[r])
            g = seq_stride(int(self.start[r]), int(self.end[r]), self.a0_len[lo:hi], self.a1_len[lo:hi])
            lo = int(self.q_off[r]); hi = lo + int(self.q_cnt[r])
            g2 = seq_stride(int(self.start[r]), int(self.end[r]), self.a0_len[lo:hi], self.a1_len[lo:hi])
            out[r] = max(g, g2, 1)
        return out

    def slice(self, lo, hi):
        """Regions [lo, hi) as a new batch sharing the variant arrays."""
        s = np.s_[lo:hi]
        return RegionBatch(self.region_id[s], self.contig_idx[s], self.start[s], self.end[s],
                           self.t_off[s], self.t_cnt[s], self.q_off[s], self.q_cnt[s],
                           self.var_pos, self.var_type, self.var_zyg, self.var_raw_space,
                           self.a0_off, self.a0_len, self.a1_off, self.a1_len, self.allele_bytes)


class CompactBatch:
    """The same batch in the library's compact form (avk_compact_batch): 20 bytes per region, 17 per call.  `from_region_batch` checks the constraints the
    narrow fields rest on (query calls of a region right behind its truth calls, allele1 right behind allele0, 32-bit positions and offsets)."""

    FIELDS = ("contig_idx", "start", "len", "v_off", "t_cnt", "q_cnt", "var_pos", "var_type_zyg", "a_off", "a0_len", "a1_len", "var_raw_space", "allele_bytes")
    DTYPES = (np.uint32, np.uint32, np.uint32, np.uint32, np.uint16, np.uint16, np.uint32, np.uint8, np.uint32, np.uint32, np.uint32, np.uint32, np.uint8)

    def __init__(self, **arrays):
        for name, dt in zip(self.FIELDS, self.DTYPES):
            a = arrays.get(name)
            setattr(self, name, None if a is None else np.ascontiguousarray(a, dtype=dt))
        self.n_regions = int(self.start.size)
        self.n_variants = int(self.var_pos.size)

    @classmethod
    def from_region_batch(cls, b, keep_raw_space=None):
        ok = (np.array_equal(b.q_off, b.t_off + b.t_cnt) and np.array_equal(b.a1_off, b.a0_off + b.a0_len) and
              (b.n_regions == 0 or (int(b.end.max()) < 2 ** 32 and int(b.t_cnt.max()) < 65536 and int(b.q_cnt.max()) < 65536)) and
              b.n_variants < 2 ** 32 and b.allele_bytes.size < 2 ** 32 and (b.n_variants == 0 or int(b.var_pos.max()) < 2 ** 32) and bool(np.all(b.end >= b.start)) and
              (b.n_variants == 0 or (int(b.var_type.max()) < 16 and int(b.var_zyg.max()) < 16)))
        if not ok:
            raise ValueError("the batch does not satisfy the constraints of the compact form (include/aardvark_amd.h: avk_compact_batch)")
        default_raw = np.array_equal(b.var_raw_space, np.maximum(b.a0_len, b.a1_len))
        if keep_raw_space is None:
            keep_raw_space = not default_raw
        return cls(contig_idx=b.contig_idx, start=b.start, len=b.end - b.start, v_off=b.t_off, t_cnt=b.t_cnt, q_cnt=b.q_cnt, var_pos=b.var_pos,
                   var_type_zyg=b.var_type | (b.var_zyg << 4), a_off=b.a0_off, a0_len=b.a0_len, a1_len=b.a1_len, var_raw_space=b.var_raw_space if keep_raw_space else None,
                   allele_bytes=b.allele_bytes)

    def widen(self):
        """back to the wide form (what the library's dp_widen kernel writes on the device)"""
        raw = self.var_raw_space if self.var_raw_space is not None else np.maximum(self.a0_len, self.a1_len)
        toff = self.v_off.astype(np.uint64)
        return RegionBatch(np.arange(self.n_regions), self.contig_idx, self.start, self.start.astype(np.uint64) + self.len, toff, self.t_cnt, toff + self.t_cnt, self.q_cnt, self.var_pos,
                           self.var_type_zyg & 15, self.var_type_zyg >> 4, raw, self.a_off, self.a0_len, self.a_off.astype(np.uint64) + self.a0_len, self.a1_len, self.allele_bytes)

    def nbytes(self):
        return sum(getattr(self, f).nbytes for f in self.FIELDS if getattr(self, f) is not None)

    def c_struct(self):
        b = AvkCompactBatch()
        b.n_regions, b.n_variants = self.n_regions, self.n_variants
        for name, ct in (("contig_idx", C.c_uint32), ("start", C.c_uint32), ("len", C.c_uint32), ("v_off", C.c_uint32), ("t_cnt", C.c_uint16), ("q_cnt", C.c_uint16), ("var_pos", C.c_uint32),
                         ("var_type_zyg", C.c_uint8), ("a_off", C.c_uint32), ("a0_len", C.c_uint32), ("a1_len", C.c_uint32), ("var_raw_space", C.c_uint32), ("allele_bytes", C.c_uint8)):
            a = getattr(self, name)
            if a is not None:
                setattr(b, name, _ptr(a, ct))
        b.allele_bytes_len = int(self.allele_bytes.size)
        return b


class PackedBatch:
    """The same batch in the library's packed form (avk_packed_batch): 10 bytes per region, 5 per call plus the allele bytes; every offset is implied by
    order.  `from_compact` checks the constraints (calls and alleles back to back in region / call order, narrow fields wide enough)."""

    FIELDS = ("contig_idx", "start", "len", "t_cnt", "q_cnt", "var_rel_pos", "var_type_zyg", "a0_len", "a1_len", "var_raw_space", "allele_bytes")
    DTYPES = (np.uint16, np.uint32, np.uint16, np.uint8, np.uint8, np.uint16, np.uint8, np.uint8, np.uint8, np.uint32, np.uint8)

    def __init__(self, **arrays):
        for name, dt in zip(self.FIELDS, self.DTYPES):
            a = arrays.get(name)
            setattr(self, name, None if a is None else np.ascontiguousarray(a, dtype=dt))
        self.n_regions = int(self.start.size)
        self.n_variants = int(self.var_rel_pos.size)

    @classmethod
    def from_compact(cls, cb):
        n, nv = cb.n_regions, cb.n_variants
        cnt = cb.t_cnt.astype(np.int64) + cb.q_cnt
        voff = np.concatenate([[0], np.cumsum(cnt)])
        alen = cb.a0_len.astype(np.int64) + cb.a1_len
        aoff = np.concatenate([[0], np.cumsum(alen)])
        ok = (np.array_equal(cb.v_off, voff[:-1]) and int(voff[-1]) == nv and np.array_equal(cb.a_off, aoff[:-1]) and (nv == 0 or int(aoff[-1]) == cb.allele_bytes.size) and
              (n == 0 or (int(cb.len.max()) < 65536 and int(cb.t_cnt.max()) < 256 and int(cb.q_cnt.max()) < 256)) and
              (nv == 0 or (int(cb.a0_len.max()) < 256 and int(cb.a1_len.max()) < 256)) and (cb.contig_idx is None or n == 0 or int(cb.contig_idx.max()) < 65536))
        rel = None
        if ok:
            region_of = np.repeat(np.arange(n), cnt)
            rel = cb.var_pos.astype(np.int64) - cb.start.astype(np.int64)[region_of]
            ok = nv == 0 or (int(rel.min()) >= 0 and int(rel.max()) < 65536)
        if not ok:
            raise ValueError("the batch does not satisfy the constraints of the packed form (include/aardvark_amd.h: avk_packed_batch)")
        return cls(contig_idx=cb.contig_idx, start=cb.start, len=cb.len, t_cnt=cb.t_cnt, q_cnt=cb.q_cnt, var_rel_pos=rel, var_type_zyg=cb.var_type_zyg, a0_len=cb.a0_len,
                   a1_len=cb.a1_len, var_raw_space=cb.var_raw_space, allele_bytes=cb.allele_bytes)

    def nbytes(self):
        return sum(getattr(self, f).nbytes for f in self.FIELDS if getattr(self, f) is not None)

    def split(self, n_parts):
        """the batch as `n_parts` batches of consecutive regions (cut at contig boundaries where there are enough contigs): a job handed over piece by piece"""
        n = self.n_regions
        cuts = [n * i // n_parts for i in range(n_parts + 1)]
        if self.contig_idx is not None and n:
            starts = np.flatnonzero(np.diff(self.contig_idx.astype(np.int64)) != 0) + 1
            if starts.size + 1 >= n_parts:
                for i in range(1, n_parts):
                    cuts[i] = int(starts[np.argmin(np.abs(starts - cuts[i]))])
        cuts = sorted(set(cuts))
        voff = np.concatenate([[0], np.cumsum(self.t_cnt.astype(np.int64) + self.q_cnt)])
        aoff = np.concatenate([[0], np.cumsum(self.a0_len.astype(np.int64) + self.a1_len)])
        parts = []
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            v0, v1 = int(voff[r0]), int(voff[r1])
            a0, a1 = int(aoff[v0]), int(aoff[v1])
            cut = {"contig_idx": (r0, r1), "start": (r0, r1), "len": (r0, r1), "t_cnt": (r0, r1), "q_cnt": (r0, r1), "var_rel_pos": (v0, v1), "var_type_zyg": (v0, v1),
                   "a0_len": (v0, v1), "a1_len": (v0, v1), "var_raw_space": (v0, v1), "allele_bytes": (a0, a1)}
            parts.append(PackedBatch(**{f: (None if getattr(self, f) is None else getattr(self, f)[cut[f][0]:cut[f][1]]) for f in self.FIELDS}))
        return parts

    def c_struct(self):
        b = AvkPackedBatch()
        b.n_regions, b.n_variants = self.n_regions, self.n_variants
        for name, ct in (("contig_idx", C.c_uint16), ("start", C.c_uint32), ("len", C.c_uint16), ("t_cnt", C.c_uint8), ("q_cnt", C.c_uint8), ("var_rel_pos", C.c_uint16),
                         ("var_type_zyg", C.c_uint8), ("a0_len", C.c_uint8), ("a1_len", C.c_uint8), ("var_raw_space", C.c_uint32), ("allele_bytes", C.c_uint8)):
            a = getattr(self, name)
            if a is not None:
                setattr(b, name, _ptr(a, ct))
        b.allele_bytes_len = int(self.allele_bytes.size)
        return b


class ResultBatch:
    """Caller-allocated outputs of one avk_compare_batch / orc_compare_batch call."""

    WIDE_REGION = (("status", np.int32), ("ed_h1", np.uint32), ("ed_h2", np.uint32), ("n_optima", np.uint32), ("type_present", np.uint16))
    WIDE_CALL = ("var_expected", "var_observed", "var_class", "var_zyg")

    def __init__(self, batch, sequences=False, group_metrics=True, bp_groups=False, packed=False):
        """packed: False = the wide arrays; True = the wide arrays and the packed form (avk_result_batch::region_packed / var_packed); "only" = the packed form
        alone — 8 bytes per region and 1 per call cross PCIe, the wide arrays are None until expand()"""
        n, v = batch.n_regions, batch.n_variants
        self.n_regions, self.n_variants = n, v
        # compact per-region BASEPAIR groups (avk_result_batch::bp_off / bp_groups); bp_groups="packed": one word per region, the groups of the regions that need
        # more than a word spilled into bp_groups (avk_result_batch::bp_packed / bp_spilled)
        self.bp_off = np.zeros(n + 1, np.uint32) if bp_groups and bp_groups != "packed" else None
        self.bp_groups = np.zeros((n + v + 1, 4), np.uint32) if bp_groups else None
        self.bp_packed = np.zeros(max(n, 1), np.uint32) if bp_groups == "packed" else None
        self.bp_spilled = np.zeros(1, np.uint32) if bp_groups == "packed" else None
        wide = packed != "only"
        self.status = np.full(n, -1, np.int32) if wide else None
        self.ed_h1 = np.zeros(n, np.uint32) if wide else None
        self.ed_h2 = np.zeros(n, np.uint32) if wide else None
        self.n_optima = np.zeros(n, np.uint32) if wide else None
        self.type_present = np.zeros(n, np.uint16) if wide else None
        self.group_metrics = np.zeros((n, N_GROUPS, N_FIELDS), np.uint32) if group_metrics else None
        for f in self.WIDE_CALL:
            setattr(self, f, np.zeros(max(v, 1), np.uint8) if wide else None)
        self.region_packed = np.full(max(n, 1), 0x7F, np.uint64) if packed else None
        self.var_packed = np.zeros(max(v, 1), np.uint8) if packed else None
        self.tally = np.zeros(TALLY_LEN, np.uint64)
        self.sequences = sequences
        if sequences:
            self.seq_stride = batch.seq_strides()
            self.seq_off = np.zeros(n, np.uint64)
            if n:
                self.seq_off[1:] = np.cumsum(self.seq_stride[:-1].astype(np.uint64) * 5)
            total = int((self.seq_stride.astype(np.uint64) * 5).sum())
            self.seq_bytes = np.zeros(max(total, 1), np.uint8)
            self.seq_len = np.zeros((n, 5), np.uint32)

    def c_struct(self):
        o = AvkResultBatch()
        if self.status is not None:
            o.status = _ptr(self.status, C.c_int32)
            o.ed_h1 = _ptr(self.ed_h1, C.c_uint32)
            o.ed_h2 = _ptr(self.ed_h2, C.c_uint32)
            o.n_optima = _ptr(self.n_optima, C.c_uint32)
            o.type_present = _ptr(self.type_present, C.c_uint16)
        if self.group_metrics is not None:
            o.group_metrics = _ptr(self.group_metrics, C.c_uint32)
        if self.var_expected is not None:
            o.var_expected = _ptr(self.var_expected, C.c_uint8)
            o.var_observed = _ptr(self.var_observed, C.c_uint8)
            o.var_class = _ptr(self.var_class, C.c_uint8)
            o.var_zyg = _ptr(self.var_zyg, C.c_uint8)
        if self.region_packed is not None:
            o.region_packed = _ptr(self.region_packed, C.c_uint64)
            o.var_packed = _ptr(self.var_packed, C.c_uint8)
        o.tally = _ptr(self.tally, C.c_uint64)
        if self.bp_off is not None:
            o.bp_off = _ptr(self.bp_off, C.c_uint32)
            o.bp_groups = _ptr(self.bp_groups, C.c_uint32)
        if getattr(self, "bp_packed", None) is not None:
            o.bp_packed = _ptr(self.bp_packed, C.c_uint32)
            o.bp_spilled = _ptr(self.bp_spilled, C.c_uint32)
            o.bp_groups = _ptr(self.bp_groups, C.c_uint32)
        if self.sequences:
            o.seq_bytes = _ptr(self.seq_bytes, C.c_uint8)
            o.seq_off = _ptr(self.seq_off, C.c_uint64)
            o.seq_stride = _ptr(self.seq_stride, C.c_uint32)
            o.seq_len = _ptr(self.seq_len, C.c_uint32)
        return o

    def expanded(self, lib, batch):
        """avk_results_expand: a ResultBatch with the wide arrays rebuilt from this one's packed form (host work, no GPU involved); group_metrics,
        bp_groups, tally and sequences are shared with this one"""
        import copy
        out = copy.copy(self)
        n, v = self.n_regions, self.n_variants
        for f, dt in self.WIDE_REGION:
            setattr(out, f, np.full(n, -1, dt) if f == "status" else np.zeros(n, dt))
        for f in self.WIDE_CALL:
            setattr(out, f, np.zeros(max(v, 1), np.uint8))
        cb, src, dst = batch.c_struct(), self.c_struct(), out.c_struct()
        lib.avk_results_expand.argtypes = [C.POINTER(AvkRegionBatch), C.POINTER(AvkResultBatch), C.POINTER(AvkResultBatch)]
        rc = lib.avk_results_expand(C.byref(cb), C.byref(src), C.byref(dst))
        if rc != 0:
            raise ValueError("avk_results_expand: %d" % rc)
        return out

    def sequence(self, r, k):
        off = int(self.seq_off[r]) + k * int(self.seq_stride[r])
        return bytes(self.seq_bytes[off:off + int(self.seq_len[r, k])])

    FIELDS_CMP = ["status", "ed_h1", "ed_h2", "n_optima", "type_present", "group_metrics",
                  "var_expected", "var_observed", "var_class", "var_zyg", "tally"]

    def diff(self, other):
        """Names of the output arrays that differ from `other` (bit-exact comparison)."""
        bad = []
        for f in self.FIELDS_CMP:
            a, b = getattr(self, f), getattr(other, f)
            if a is None or b is None:
                continue
            if not np.array_equal(a, b):
                bad.append(f)
        if self.sequences and other.sequences:
            if not np.array_equal(self.seq_len, other.seq_len):
                bad.append("seq_len")
            else:
                for r in range(self.seq_len.shape[0]):
                    for k in range(5):
                        if self.sequence(r, k) != other.sequence(r, k):
                            bad.append("seq_bytes[%d,%d]" % (r, k))
                            break
        return bad
