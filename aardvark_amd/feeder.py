"""ctypes plumbing for libaardvark_feeder.so (include/aardvark_feeder.h): FASTA / BED / VCF -> RegionBatch with the
reference's region ids and windows (src/parsing/region_generation.rs), and the summary.tsv writer
(src/writers/summary.rs).  Host code only; the solver stays in libaardvark_amd.so."""
import ctypes as C
import os

import numpy as np

from ._abi import AvkRegionBatch, RegionBatch

METRIC_GT, METRIC_HAP, METRIC_WEIGHTED_HAP, METRIC_BASEPAIR, METRIC_RECORD_BP = 1, 2, 4, 8, 16
_lib = None


class FeederError(RuntimeError):
    pass


def library_path():
    return os.environ.get("AVF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libaardvark_feeder.so")


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FeederError("%s is missing: build it with `python __graft_entry__.py` (make -C aardvark_amd/csrc/feeder)" % path)
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.avf_last_error.restype = C.c_char_p
    lib.avf_genome_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.avf_genome_n_contigs.restype = C.c_uint32
    lib.avf_genome_n_contigs.argtypes = [vp]
    lib.avf_genome_name.restype = C.c_char_p
    lib.avf_genome_name.argtypes = [vp, C.c_uint32]
    lib.avf_genome_seq.restype = C.POINTER(C.c_uint8)
    lib.avf_genome_seq.argtypes = [vp, C.c_uint32]
    lib.avf_genome_len.restype = C.c_uint64
    lib.avf_genome_len.argtypes = [vp, C.c_uint32]
    lib.avf_genome_free.argtypes = [vp]
    lib.avf_feed_compare.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    lib.avf_feed_batch.restype = C.POINTER(AvkRegionBatch)
    lib.avf_feed_batch.argtypes = [vp]
    lib.avf_feed_var_record.restype = C.POINTER(C.c_uint64)
    lib.avf_feed_var_record.argtypes = [vp]
    lib.avf_feed_var_alt_index.restype = C.POINTER(C.c_uint32)
    lib.avf_feed_var_alt_index.argtypes = [vp]
    lib.avf_feed_loaded_variants.restype = C.c_uint64
    lib.avf_feed_loaded_variants.argtypes = [vp, C.c_int]
    lib.avf_feed_free.argtypes = [vp]
    lib.avf_write_summary.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint32]
    u8p = C.POINTER(C.c_uint8)
    lib.avf_write_annotated_vcf.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(AvkRegionBatch), C.c_int,
                                            C.POINTER(C.c_int32), u8p, u8p, u8p]
    lib.avf_strat_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.avf_strat_n_labels.restype = C.c_uint32
    lib.avf_strat_n_labels.argtypes = [vp]
    lib.avf_strat_label.restype = C.c_char_p
    lib.avf_strat_label.argtypes = [vp, C.c_uint32]
    lib.avf_strat_n_intervals.restype = C.c_uint64
    lib.avf_strat_n_intervals.argtypes = [vp, C.c_uint32, C.c_char_p]
    u32p = C.POINTER(C.c_uint32)
    for f in (lib.avf_strat_containments, lib.avf_strat_overlaps):
        f.restype = C.c_uint32
        f.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int64, u32p, C.c_uint32]
    lib.avf_strat_region_labels.restype = C.c_uint32
    lib.avf_strat_region_labels.argtypes = [vp, vp, C.POINTER(AvkRegionBatch), C.c_uint64, u32p, C.c_uint32]
    lib.avf_strat_free.argtypes = [vp]
    lib.avf_strat_batch_labels.argtypes = [vp, vp, C.POINTER(AvkRegionBatch), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), u32p]
    lib.avf_write_summary_stratified.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), vp, C.POINTER(C.c_uint64), C.c_uint32]
    lib.avf_region_summary_open.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(vp)]
    lib.avf_region_summary_rows.argtypes = [vp, vp, C.POINTER(AvkRegionBatch), C.c_uint64, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    lib.avf_region_sequences_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.avf_region_sequences_rows.argtypes = [vp, vp, C.POINTER(AvkRegionBatch), C.c_uint64, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.avf_table_close.argtypes = [vp]
    from .merge import AvkMultiBatch
    cpp = C.POINTER(C.c_char_p)
    lib.avf_feed_merge.argtypes = [C.c_uint32, cpp, cpp, C.c_char_p, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    lib.avf_feed_multi_batch.restype = C.POINTER(AvkMultiBatch)
    lib.avf_feed_multi_batch.argtypes = [vp]
    lib.avf_write_merge_outputs.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, vp, C.POINTER(AvkMultiBatch), cpp,
                                            C.POINTER(C.c_int32), u8p, C.POINTER(C.c_uint64)]
    lib.avf_write_merge_summary.argtypes = [C.c_char_p, C.POINTER(AvkMultiBatch), cpp, C.POINTER(C.c_int32), u8p, C.POINTER(C.c_uint64)]
    lib.avf_write_merge_summary_counts.argtypes = [C.c_char_p, C.c_uint32, cpp, C.POINTER(C.c_uint64), C.c_uint64]
    lib.avf_calls_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.avf_calls_free.argtypes = [vp]
    lib.avf_feed_from_calls.argtypes = [C.c_uint32, C.POINTER(vp), C.c_char_p, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    lib.avf_vcf_sample_name.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64]
    _lib = lib
    return lib


def _check(lib, rc):
    if rc:
        raise FeederError((lib.avf_last_error() or b"").decode(errors="replace"))


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Genome:
    """ReferenceGenome::from_fasta: contigs in file order.  case: "upper" (default: soft-masked bases are loaded as upper case, like
    the tools' --reference-case upper) or "raw" (the file's bytes); see avf_genome_load_case in include/aardvark_feeder.h."""

    def __init__(self, fasta_path, case="upper"):
        if case not in ("upper", "raw"):
            raise ValueError("case must be 'upper' or 'raw'")
        self.lib = load_library()
        h = C.c_void_p()
        self.lib.avf_genome_load_case.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        _check(self.lib, self.lib.avf_genome_load_case(os.fsencode(fasta_path), 1 if case == "upper" else 0, C.byref(h)))
        self.handle = h
        n = self.lib.avf_genome_n_contigs(h)
        self.names = [self.lib.avf_genome_name(h, i).decode() for i in range(n)]

    def contigs(self):
        """list of numpy uint8 arrays (copies), in file order: what avk_ref_upload takes"""
        return [_arr(self.lib.avf_genome_seq(self.handle, i), int(self.lib.avf_genome_len(self.handle, i)), np.uint8) for i in range(len(self.names))]

    def close(self):
        if self.handle:
            self.lib.avf_genome_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Feed:
    """The result of RegionIterator over a truth/query VCF pair: a RegionBatch plus provenance; `packed` = the same batch in the packed form
    (avf_feed_pack -> PackedBatch; merge feeds: avf_feed_pack_multi -> merge.PackedMultiBatch), None when the call sets do not fit that form."""

    def __init__(self, batch, var_record, var_alt_index, loaded, packed=None):
        self.batch, self.var_record, self.var_alt_index, self.loaded, self.packed = batch, var_record, var_alt_index, loaded, packed


_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


def _take_packed_multi(lib, h):
    """avf_feed_pack_multi into numpy-owned buffers -> merge.PackedMultiBatch, or None when the feed does not fit the form"""
    from .merge import AvkPackedMultiBatch, PackedMultiBatch
    bufs = {}

    def alloc(_user, nbytes):
        a = np.empty(max(int(nbytes), 1), np.uint8)
        bufs[a.ctypes.data] = a
        return a.ctypes.data

    cb = _ALLOC(alloc)
    out = AvkPackedMultiBatch()
    lib.avf_feed_pack_multi.argtypes = [C.c_void_p, _ALLOC, C.c_void_p, C.POINTER(AvkPackedMultiBatch)]
    rc = lib.avf_feed_pack_multi(h, cb, None, C.byref(out))
    if rc == 1:
        return None
    _check(lib, rc)
    n, nv, na, k = int(out.n_regions), int(out.n_variants), int(out.allele_bytes_len), int(out.n_inputs)

    def view(ptr, count, dtype):
        addr = C.cast(ptr, C.c_void_p).value
        if addr is None:
            return None
        return bufs[addr][:count * np.dtype(dtype).itemsize].view(dtype)

    return PackedMultiBatch(k, contig_idx=view(out.contig_idx, n, np.uint16), start=view(out.start, n, np.uint32), len=view(out.len, n, np.uint16),
                            in_cnt=view(out.in_cnt, n * k, np.uint8), var_rel_pos=view(out.var_rel_pos, nv, np.uint16), var_type_zyg=view(out.var_type_zyg, nv, np.uint8),
                            a0_len=view(out.a0_len, nv, np.uint8), a1_len=view(out.a1_len, nv, np.uint8), var_raw_space=view(out.var_raw_space, nv, np.uint32),
                            allele_bytes=view(out.allele_bytes, max(na, 1), np.uint8))


def _take_packed(lib, h):
    """avf_feed_pack into numpy-owned buffers -> PackedBatch, or None when the feed does not fit the form"""
    from ._abi import AvkPackedBatch, PackedBatch
    bufs = {}

    def alloc(_user, nbytes):
        a = np.empty(max(int(nbytes), 1), np.uint8)
        bufs[a.ctypes.data] = a
        return a.ctypes.data

    cb = _ALLOC(alloc)
    out = AvkPackedBatch()
    lib.avf_feed_pack.argtypes = [C.c_void_p, _ALLOC, C.c_void_p, C.POINTER(AvkPackedBatch)]
    rc = lib.avf_feed_pack(h, cb, None, C.byref(out))
    if rc == 1:
        return None
    _check(lib, rc)
    n, nv, na = int(out.n_regions), int(out.n_variants), int(out.allele_bytes_len)

    def view(ptr, count, dtype):
        addr = C.cast(ptr, C.c_void_p).value
        if addr is None:
            return None
        return bufs[addr][:count * np.dtype(dtype).itemsize].view(dtype)

    return PackedBatch(contig_idx=view(out.contig_idx, n, np.uint16), start=view(out.start, n, np.uint32), len=view(out.len, n, np.uint16),
                       t_cnt=view(out.t_cnt, n, np.uint8), q_cnt=view(out.q_cnt, n, np.uint8), var_rel_pos=view(out.var_rel_pos, nv, np.uint16),
                       var_type_zyg=view(out.var_type_zyg, nv, np.uint8), a0_len=view(out.a0_len, nv, np.uint8), a1_len=view(out.a1_len, nv, np.uint8),
                       var_raw_space=view(out.var_raw_space, nv, np.uint32), allele_bytes=view(out.allele_bytes, na, np.uint8))


def vcf_sample_name(vcf, index=0):
    """get_vcf_sample_name: the name of sample `index` of the file's #CHROM line"""
    lib = load_library()
    buf = C.create_string_buffer(4096)
    _check(lib, lib.avf_vcf_sample_name(os.fsencode(vcf), index, buf, len(buf)))
    return buf.value.decode()


class Calls:
    """One VCF's calls, every chromosome (avf_calls_load): the first half of a feed, loadable while the genome is still loading."""

    def __init__(self, vcf, sample="", enable_trimming=True):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.avf_calls_load(os.fsencode(vcf), sample.encode(), 1 if enable_trimming else 0, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.avf_calls_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def feed_from_calls(calls, regions_bed, genome, min_variant_gap=50, merge=False):
    """avf_feed_from_calls: the region walk over loaded call sets -> the Feed of feed_compare (two inputs) or feed_merge"""
    lib = load_library()
    h = C.c_void_p()
    arr = (C.c_void_p * len(calls))(*[c.handle for c in calls])
    _check(lib, lib.avf_feed_from_calls(len(calls), arr, os.fsencode(regions_bed) if regions_bed else None, genome.handle, min_variant_gap, 1 if merge else 0, C.byref(h)))
    return _take_feed(lib, h, len(calls), merge)


def _take_feed(lib, h, k, merge):
    try:
        if merge:
            from .merge import MultiBatch
            b = lib.avf_feed_multi_batch(h).contents
            n, nv = int(b.n_regions), int(b.n_variants)
            batch = MultiBatch(k, region_id=_arr(b.region_id, n, np.uint64), contig_idx=_arr(b.contig_idx, n, np.uint32), start=_arr(b.start, n, np.uint64),
                               end=_arr(b.end, n, np.uint64), in_off=_arr(b.in_off, n * k, np.uint64), in_cnt=_arr(b.in_cnt, n * k, np.uint32),
                               var_pos=_arr(b.var_pos, nv, np.uint64), var_type=_arr(b.var_type, nv, np.uint8), var_zyg=_arr(b.var_zyg, nv, np.uint8),
                               var_raw_space=_arr(b.var_raw_space, nv, np.uint32), a0_off=_arr(b.a0_off, nv, np.uint64), a0_len=_arr(b.a0_len, nv, np.uint32),
                               a1_off=_arr(b.a1_off, nv, np.uint64), a1_len=_arr(b.a1_len, nv, np.uint32),
                               allele_bytes=_arr(b.allele_bytes, int(b.allele_bytes_len), np.uint8))
        else:
            b = lib.avf_feed_batch(h).contents
            n, nv = int(b.n_regions), int(b.n_variants)
            batch = RegionBatch(_arr(b.region_id, n, np.uint64), _arr(b.contig_idx, n, np.uint32), _arr(b.start, n, np.uint64), _arr(b.end, n, np.uint64),
                                _arr(b.t_off, n, np.uint64), _arr(b.t_cnt, n, np.uint32), _arr(b.q_off, n, np.uint64), _arr(b.q_cnt, n, np.uint32),
                                _arr(b.var_pos, nv, np.uint64), _arr(b.var_type, nv, np.uint8), _arr(b.var_zyg, nv, np.uint8), _arr(b.var_raw_space, nv, np.uint32),
                                _arr(b.a0_off, nv, np.uint64), _arr(b.a0_len, nv, np.uint32), _arr(b.a1_off, nv, np.uint64), _arr(b.a1_len, nv, np.uint32),
                                _arr(b.allele_bytes, int(b.allele_bytes_len), np.uint8))
        return Feed(batch, _arr(lib.avf_feed_var_record(h), nv, np.uint64), _arr(lib.avf_feed_var_alt_index(h), nv, np.uint32),
                    tuple(int(lib.avf_feed_loaded_variants(h, i)) for i in range(k)), _take_packed_multi(lib, h) if merge else _take_packed(lib, h))
    finally:
        lib.avf_feed_free(h)


def feed_compare(truth_vcf, query_vcf, regions_bed, genome, truth_sample="", query_sample="", min_variant_gap=50, enable_trimming=True):
    lib = load_library()
    h = C.c_void_p()
    _check(lib, lib.avf_feed_compare(os.fsencode(truth_vcf), truth_sample.encode(), os.fsencode(query_vcf), query_sample.encode(),
                                     os.fsencode(regions_bed) if regions_bed else None, genome.handle, min_variant_gap, 1 if enable_trimming else 0, C.byref(h)))
    try:
        b = lib.avf_feed_batch(h).contents
        n, nv = int(b.n_regions), int(b.n_variants)
        batch = RegionBatch(_arr(b.region_id, n, np.uint64), _arr(b.contig_idx, n, np.uint32), _arr(b.start, n, np.uint64), _arr(b.end, n, np.uint64),
                            _arr(b.t_off, n, np.uint64), _arr(b.t_cnt, n, np.uint32), _arr(b.q_off, n, np.uint64), _arr(b.q_cnt, n, np.uint32),
                            _arr(b.var_pos, nv, np.uint64), _arr(b.var_type, nv, np.uint8), _arr(b.var_zyg, nv, np.uint8), _arr(b.var_raw_space, nv, np.uint32),
                            _arr(b.a0_off, nv, np.uint64), _arr(b.a0_len, nv, np.uint32), _arr(b.a1_off, nv, np.uint64), _arr(b.a1_len, nv, np.uint32),
                            _arr(b.allele_bytes, int(b.allele_bytes_len), np.uint8))
        return Feed(batch, _arr(lib.avf_feed_var_record(h), nv, np.uint64), _arr(lib.avf_feed_var_alt_index(h), nv, np.uint32),
                    (int(lib.avf_feed_loaded_variants(h, 0)), int(lib.avf_feed_loaded_variants(h, 1))), _take_packed(lib, h))
    finally:
        lib.avf_feed_free(h)


def write_summary(path, tally, compare_label="compare", metrics=METRIC_GT | METRIC_BASEPAIR):
    lib = load_library()
    t = np.ascontiguousarray(tally, dtype=np.uint64)
    _check(lib, lib.avf_write_summary(os.fsencode(path), compare_label.encode(), t.ctypes.data_as(C.POINTER(C.c_uint64)), metrics))


def write_annotated_vcf(out_path, input_vcf, genome, batch, result, source, sample_name="", version="aardvark_amd", command_line=""):
    """truth.vcf.gz (source 0) or query.vcf.gz (source 1) of `compare` + its .tbi; `result` is the ResultBatch of `batch`."""
    lib = load_library()
    cb = batch.c_struct()
    u8 = lambda a: np.ascontiguousarray(a, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
    st = np.ascontiguousarray(result.status, np.int32)
    keep = [np.ascontiguousarray(x, np.uint8) for x in (result.var_expected, result.var_observed, result.var_class)]
    _check(lib, lib.avf_write_annotated_vcf(os.fsencode(out_path), os.fsencode(input_vcf), sample_name.encode(), version.encode(), command_line.encode(),
                                            genome.handle, C.byref(cb), source, st.ctypes.data_as(C.POINTER(C.c_int32)), u8(keep[0]), u8(keep[1]), u8(keep[2])))


class Stratifications:
    """Stratifications::from_tsv_batch (src/parsing/stratifications.rs): labelled BED sets, labels in sorted order."""

    def __init__(self, tsv_path):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.avf_strat_load(os.fsencode(tsv_path), C.byref(h)))
        self.handle = h
        self.labels = [self.lib.avf_strat_label(h, i).decode() for i in range(self.lib.avf_strat_n_labels(h))]

    def _query(self, fn, chrom, first, last):
        out = (C.c_uint32 * max(len(self.labels), 1))()
        n = fn(self.handle, chrom.encode(), first, last, out, len(self.labels))
        return [int(out[i]) for i in range(n)]

    def containments(self, chrom, first, last):
        return self._query(self.lib.avf_strat_containments, chrom, first, last)

    def overlaps(self, chrom, first, last):
        return self._query(self.lib.avf_strat_overlaps, chrom, first, last)

    def n_intervals(self, label, chrom):
        return int(self.lib.avf_strat_n_intervals(self.handle, label, chrom.encode()))

    def region_labels(self, genome, batch, r):
        cb = batch.c_struct()
        out = (C.c_uint32 * max(len(self.labels), 1))()
        n = self.lib.avf_strat_region_labels(self.handle, genome.handle, C.byref(cb), r, out, len(self.labels))
        return [int(out[i]) for i in range(n)]

    def batch_labels(self, genome, batch, first=0, n=None):
        """avf_strat_batch_labels: (label_off[n + 1], label_idx) — the labels of region first + k are label_idx[label_off[k]:label_off[k + 1]]"""
        n = batch.n_regions - first if n is None else n
        cb = batch.c_struct()
        off = np.zeros(n + 1, np.uint64)
        u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        _check(self.lib, self.lib.avf_strat_batch_labels(self.handle, genome.handle, C.byref(cb), first, n, off.ctypes.data_as(u64p), None))
        idx = np.zeros(int(off[n]) + 1, np.uint32)
        _check(self.lib, self.lib.avf_strat_batch_labels(self.handle, genome.handle, C.byref(cb), first, n, off.ctypes.data_as(u64p), idx.ctypes.data_as(u32p)))
        return off, idx[:int(off[n])]

    def close(self):
        if self.handle:
            self.lib.avf_strat_free(self.handle)
            self.handle = None


def write_summary_stratified(path, tally, strat, strat_tallies, compare_label="compare", metrics=METRIC_GT | METRIC_BASEPAIR):
    lib = load_library()
    t = np.ascontiguousarray(tally, dtype=np.uint64)
    st = np.ascontiguousarray(strat_tallies, dtype=np.uint64).reshape(-1)
    _check(lib, lib.avf_write_summary_stratified(os.fsencode(path), compare_label.encode(), t.ctypes.data_as(C.POINTER(C.c_uint64)), strat.handle,
                                                 st.ctypes.data_as(C.POINTER(C.c_uint64)), metrics))


def write_debug_tables(summary_path, sequences_path, genome, batch, result, metrics=METRIC_GT | METRIC_BASEPAIR):
    """region_summary.tsv.gz and region_sequences.tsv.gz of --output-debug for one batch and its ResultBatch (with sequences)."""
    lib = load_library()
    cb = batch.c_struct()
    st = np.ascontiguousarray(result.status, np.int32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for path, is_seq in ((summary_path, False), (sequences_path, True)):
        if not path:
            continue
        h = C.c_void_p()
        if is_seq:
            _check(lib, lib.avf_region_sequences_open(os.fsencode(path), C.byref(h)))
            rc = lib.avf_region_sequences_rows(h, genome.handle, C.byref(cb), 0, batch.n_regions, P(st, C.c_int32), P(result.seq_bytes, C.c_uint8),
                                               P(np.ascontiguousarray(result.seq_len, np.uint32), C.c_uint32), P(result.seq_off, C.c_uint64), P(result.seq_stride, C.c_uint32))
        else:
            _check(lib, lib.avf_region_summary_open(os.fsencode(path), metrics, C.byref(h)))
            gm = np.ascontiguousarray(result.group_metrics, np.uint32)
            rc = lib.avf_region_summary_rows(h, genome.handle, C.byref(cb), 0, batch.n_regions, P(st, C.c_int32), P(gm, C.c_uint32))
        rc2 = lib.avf_table_close(h)
        _check(lib, rc or rc2)


# ---- merge (src/main.rs run_merge: region generation over k VCFs, VariantMerger, MergeSummaryWriter) -----------------------------------
def _strs(items):
    arr = (C.c_char_p * len(items))(*[os.fsencode(x) if x is not None else None for x in items])
    return arr


def feed_merge(vcfs, regions_bed, genome, samples=None, min_variant_gap=50, enable_trimming=True):
    """RegionIterator::new_merge_iterator + the iterator: a merge.MultiBatch plus provenance (Feed.batch is the MultiBatch)."""
    from .merge import MultiBatch
    lib = load_library()
    h = C.c_void_p()
    k = len(vcfs)
    samples = list(samples or []) + [""] * (k - len(samples or []))
    _check(lib, lib.avf_feed_merge(k, _strs(vcfs), _strs(samples), os.fsencode(regions_bed) if regions_bed else None, genome.handle, min_variant_gap,
                                   1 if enable_trimming else 0, C.byref(h)))
    try:
        b = lib.avf_feed_multi_batch(h).contents
        n, nv = int(b.n_regions), int(b.n_variants)
        mb = MultiBatch(k, region_id=_arr(b.region_id, n, np.uint64), contig_idx=_arr(b.contig_idx, n, np.uint32), start=_arr(b.start, n, np.uint64),
                        end=_arr(b.end, n, np.uint64), in_off=_arr(b.in_off, n * k, np.uint64), in_cnt=_arr(b.in_cnt, n * k, np.uint32),
                        var_pos=_arr(b.var_pos, nv, np.uint64), var_type=_arr(b.var_type, nv, np.uint8), var_zyg=_arr(b.var_zyg, nv, np.uint8),
                        var_raw_space=_arr(b.var_raw_space, nv, np.uint32), a0_off=_arr(b.a0_off, nv, np.uint64), a0_len=_arr(b.a0_len, nv, np.uint32),
                        a1_off=_arr(b.a1_off, nv, np.uint64), a1_len=_arr(b.a1_len, nv, np.uint32),
                        allele_bytes=_arr(b.allele_bytes, int(b.allele_bytes_len), np.uint8))
        return Feed(mb, _arr(lib.avf_feed_var_record(h), nv, np.uint64), _arr(lib.avf_feed_var_alt_index(h), nv, np.uint32),
                    tuple(int(lib.avf_feed_loaded_variants(h, i)) for i in range(k)), _take_packed_multi(lib, h))
    finally:
        lib.avf_feed_free(h)


def _merge_args(mb, result):
    st = np.ascontiguousarray(result.status, np.int32)
    cls = np.ascontiguousarray(result.classification, np.uint8)
    mem = np.ascontiguousarray(result.members, np.uint64)
    if st.size == 0:
        st, cls, mem = np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(1, np.uint64)
    return st, cls, mem


def write_merge_outputs(out_folder, primary_vcf, genome, mb, result, tags=None, sample_name="", version="aardvark_amd", command_line=""):
    """passing.vcf.gz, regions.bed.gz, failed_regions.bed.gz (+ .tbi each) of `merge`; result = merge.MergeResult of `mb`"""
    lib = load_library()
    tags = list(tags) if tags else ["vcf_%d" % i for i in range(mb.n_inputs)]
    st, cls, mem = _merge_args(mb, result)
    cb = mb.c_struct()
    _check(lib, lib.avf_write_merge_outputs(os.fsencode(out_folder), os.fsencode(primary_vcf), sample_name.encode(), version.encode(), command_line.encode(),
                                            genome.handle, C.byref(cb), _strs(tags), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                            cls.ctypes.data_as(C.POINTER(C.c_uint8)), mem.ctypes.data_as(C.POINTER(C.c_uint64))))


def write_merge_summary(path, mb, result, tags=None):
    lib = load_library()
    tags = list(tags) if tags else ["vcf_%d" % i for i in range(mb.n_inputs)]
    st, cls, mem = _merge_args(mb, result)
    cb = mb.c_struct()
    _check(lib, lib.avf_write_merge_summary(os.fsencode(path), C.byref(cb), _strs(tags), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                            cls.ctypes.data_as(C.POINTER(C.c_uint8)), mem.ctypes.data_as(C.POINTER(C.c_uint64))))


def write_merge_summary_counts(path, n_inputs, counts, tags=None):
    """the merge summary table from the dense block of sums of a sharded merge (aardvark_amd.merge.merge_counts, summed over the ranks)"""
    lib = load_library()
    tags = list(tags) if tags else ["vcf_%d" % i for i in range(n_inputs)]
    counts = np.ascontiguousarray(counts, np.uint64)
    _check(lib, lib.avf_write_merge_summary_counts(os.fsencode(path), n_inputs, _strs(tags), counts.ctypes.data_as(C.POINTER(C.c_uint64)), counts.size))
