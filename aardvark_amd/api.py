"""ctypes bindings of libaardvark_amd.so and the host-side mirror of the reference interface.

`Context.solve_compare_regions(batch, config)` is the batched form of the reference's
`solve_compare_region(problem, reference_genome, compare_config, stratifications)`
(src/waffle_solver.rs:122-124): same inputs (CompareRegion fields, CompareConfig fields), same
outputs (CompareBenchmark fields), same per-region error behaviour (a failed region yields a
status instead of metrics and the batch continues, src/main.rs:255-265).
"""
import ctypes as C
import weakref
import os

import numpy as np

from ._abi import (TALLY_LEN, AvkCompactBatch, AvkCompareConfig, AvkPackedBatch, AvkRegionBatch, AvkResultBatch, CompactBatch, PackedBatch, RegionBatch,
                   ResultBatch)

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


class AardvarkAmdError(RuntimeError):
    pass


def library_path():
    """in-tree libaardvark_amd.so; AVK_LIB selects another in-tree build (kernel tuning experiments)"""
    return os.path.join(_HERE, os.environ.get("AVK_LIB", "libaardvark_amd.so"))


def load_library():
    """Loads libaardvark_amd.so (built in-tree by __graft_entry__.build() / csrc/Makefile).
    Fails loudly when it is missing: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise AardvarkAmdError("%s not found: build it with `make -C aardvark_amd/csrc` (hipcc, gfx950)" % path)
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.avk_version.restype = C.c_char_p
    lib.avk_source_hash.restype = C.c_char_p
    lib.avk_merge_classify.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                       vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    lib.avk_edit_distance.restype = C.c_uint64
    lib.avk_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    lib.avk_last_error.restype = C.c_char_p
    lib.avk_last_error.argtypes = [vp]
    lib.avk_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.avk_ctx_destroy.argtypes = [vp]
    lib.avk_ctx_destroy.restype = None
    lib.avk_ctx_set_stream.argtypes = [vp, vp]
    lib.avk_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.avk_ref_upload.argtypes = [vp, C.c_uint32, C.POINTER(u8p), u64p]
    lib.avk_compare_batch.argtypes = [vp, C.POINTER(AvkRegionBatch), C.POINTER(AvkCompareConfig), C.POINTER(AvkResultBatch)]
    lib.avk_batch_upload.argtypes = [vp, C.POINTER(AvkRegionBatch), C.POINTER(vp)]
    lib.avk_compare_compact.argtypes = [vp, C.POINTER(AvkCompactBatch), C.POINTER(AvkCompareConfig), C.POINTER(AvkResultBatch)]
    lib.avk_batch_upload_compact.argtypes = [vp, C.POINTER(AvkCompactBatch), C.POINTER(vp)]
    lib.avk_compare_packed.argtypes = [vp, C.POINTER(AvkPackedBatch), C.POINTER(AvkCompareConfig), C.POINTER(AvkResultBatch)]
    if hasattr(lib, "avk_compare_packed_submit"):  # (AVK_LIB may name an older in-tree build: kernel A/B runs)
        lib.avk_compare_packed_submit.argtypes = [vp, C.POINTER(AvkPackedBatch), C.POINTER(AvkCompareConfig), C.POINTER(AvkResultBatch), C.POINTER(vp)]
        lib.avk_wait.argtypes = [vp, vp]
    lib.avk_batch_upload_packed.argtypes = [vp, C.POINTER(AvkPackedBatch), C.POINTER(vp)]
    lib.avk_compare_resident.argtypes = [vp, vp, C.POINTER(AvkCompareConfig), vp]
    lib.avk_results_download.argtypes = [vp, vp, C.POINTER(AvkResultBatch)]
    lib.avk_batch_free.argtypes = [vp, vp]
    lib.avk_batch_free.restype = None
    lib.avk_synchronize.argtypes = [vp]
    lib.avk_seq_stride.restype = C.c_uint32
    lib.avk_seq_stride.argtypes = [C.POINTER(AvkRegionBatch), C.c_uint64]
    lib.avk_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.avk_last_solver_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.avk_last_tier_counts.argtypes = [vp, u64p]
    lib.avk_last_compare_was_one_shot.argtypes = [vp]
    lib.avk_last_lane_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.avk_last_lane_solved.argtypes = [vp, u64p]
    lib.avk_last_wide_solved.argtypes = [vp, u64p]
    lib.avk_debug_phase_cycles.argtypes = [vp, u64p]
    lib.avk_algorithmic_bytes.restype = C.c_uint64
    lib.avk_algorithmic_bytes.argtypes = [C.POINTER(AvkRegionBatch)]
    lib.avk_algorithmic_bytes_ex.restype = C.c_uint64
    lib.avk_algorithmic_bytes_ex.argtypes = [C.POINTER(AvkRegionBatch), C.c_int]
    lib.avk_optimize_pairs_batch.argtypes = [vp, C.POINTER(AvkRegionBatch), C.c_uint32, C.POINTER(C.c_int32), u8p]
    lib.avk_group_metrics_from_compact.argtypes = [C.POINTER(AvkRegionBatch), C.c_uint64, C.POINTER(AvkResultBatch), C.POINTER(C.c_uint32)]
    lib.avk_host_alloc.restype = vp
    lib.avk_host_alloc.argtypes = [vp, C.c_size_t]
    lib.avk_host_free.restype = None
    lib.avk_host_free.argtypes = [vp, vp]
    _lib = lib
    return lib


class CompareConfig:
    """CompareConfig (reference src/waffle_solver.rs:94-115); defaults as the reference's."""

    def __init__(self, enable_sequences=True, enable_exact_shortcut=False, max_branch_factor=50):
        self.enable_sequences = enable_sequences
        self.enable_exact_shortcut = enable_exact_shortcut
        self.max_branch_factor = max_branch_factor

    def c_struct(self):
        return AvkCompareConfig(self.max_branch_factor, 1 if self.enable_sequences else 0, 1 if self.enable_exact_shortcut else 0)


class ResidentBatch:
    """A region batch living in HBM (avk_dev_batch)."""

    def __init__(self, ctx, batch):
        self.ctx, self.batch = ctx, batch
        self.handle = C.c_void_p()
        cb = batch.c_struct()
        ctx._check(ctx.lib.avk_batch_upload(ctx.handle, C.byref(cb), C.byref(self.handle)))

    def free(self):
        if self.handle:
            self.ctx.lib.avk_batch_free(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _ContextCore:
    """The native context and its pinned blocks.  Arrays handed out by Context.host_array are views of avk_host_alloc memory: each block is freed when the LAST
    array that views it is collected (a finalizer on the buffer numpy keeps as the array's base), and the native context is destroyed when it has been closed AND
    its last block is gone — so an array read after close(), or at interpreter shutdown, never touches freed memory."""

    def __init__(self, lib, handle):
        self.lib, self.handle, self.live_blocks, self.closed = lib, handle, 0, False

    def release_block(self, p):
        if self.handle:
            self.lib.avk_host_free(self.handle, p)
        self.live_blocks -= 1
        self._destroy_if_done()

    def close(self):
        self.closed = True
        self._destroy_if_done()

    def _destroy_if_done(self):
        if self.closed and self.live_blocks <= 0 and self.handle:
            self.lib.avk_ctx_destroy(self.handle)
            self.handle = C.c_void_p()


class Ticket:
    """a batch in flight (Context.submit_packed); wait() returns its ResultBatch, once"""

    def __init__(self, ctx, handle, res, keep):
        self.ctx, self.handle, self.res, self._keep = ctx, handle, res, keep

    def wait(self):
        if self.handle is None:
            raise RuntimeError("the ticket has been waited for")
        h, self.handle = self.handle, None
        self.ctx._check(self.ctx.lib.avk_wait(self.ctx.handle, h))
        self._keep = None
        return self.res


class Context:
    """One GPU context (avk_ctx): owns the uploaded reference genome and the workspaces."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.handle = C.c_void_p()
        rc = self.lib.avk_ctx_create(device, C.byref(self.handle))
        if rc != 0:
            raise AardvarkAmdError("avk_ctx_create(%d) failed (%d): %s" % (device, rc, self.lib.avk_last_error(None).decode()))
        self._contigs = None
        self._core = _ContextCore(self.lib, self.handle)

    def close(self):
        """No call may follow.  The native context goes at once, or — when arrays of host_array / pinned_* are still referenced — with the last of them."""
        if self.handle:
            self._core.close()
            self.handle = C.c_void_p()

    def host_array(self, shape, dtype):
        """a numpy array in pinned host memory (avk_host_alloc): batch and result arrays that live there are copied by DMA without a host pass;
        freed when the last array that views the block is collected (not by close(): see _ContextCore)"""
        count = int(np.prod(shape))
        nbytes = max(16, count * np.dtype(dtype).itemsize)
        p = self.lib.avk_host_alloc(self.handle, nbytes)
        if not p:
            raise AardvarkAmdError("avk_host_alloc(%d) failed: %s" % (nbytes, self.lib.avk_last_error(self.handle).decode()))
        buf = (C.c_uint8 * nbytes).from_address(p)
        self._core.live_blocks += 1
        weakref.finalize(buf, self._core.release_block, p)  # numpy keeps `buf` as the base of every view of the block
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

    def pinned_batch(self, batch):
        """a copy of `batch` whose arrays live in pinned memory"""
        def pin(a):
            out = self.host_array(a.shape, a.dtype)
            out[...] = a
            return out
        return RegionBatch(*[pin(getattr(batch, f)) for f in ("region_id", "contig_idx", "start", "end", "t_off", "t_cnt", "q_off", "q_cnt", "var_pos", "var_type", "var_zyg",
                                                              "var_raw_space", "a0_off", "a0_len", "a1_off", "a1_len", "allele_bytes")])

    def pinned_compact(self, cbatch):
        """a copy of a CompactBatch whose arrays live in pinned memory"""
        def pin(a):
            if a is None:
                return None
            out = self.host_array(a.shape, a.dtype)
            out[...] = a
            return out
        return CompactBatch(**{f: pin(getattr(cbatch, f)) for f in CompactBatch.FIELDS})

    def pinned_packed(self, pbatch):
        """a copy of a PackedBatch whose arrays live in pinned memory"""
        def pin(a):
            if a is None:
                return None
            out = self.host_array(a.shape, a.dtype)
            out[...] = a
            return out
        return PackedBatch(**{f: pin(getattr(pbatch, f)) for f in PackedBatch.FIELDS})

    def solve_packed(self, pbatch, config=None, res=None):
        """avk_compare_packed: solve_compare_region for every region of a batch in the packed form -> ResultBatch (indexed like the packed arrays)"""
        config = config or CompareConfig(enable_sequences=False)
        res = res if res is not None else ResultBatch(pbatch, sequences=False, group_metrics=False)
        pb, cfg, ro = pbatch.c_struct(), config.c_struct(), res.c_struct()
        self._check(self.lib.avk_compare_packed(self.handle, C.byref(pb), C.byref(cfg), C.byref(ro)))
        return res

    def submit_packed(self, pbatch, config=None, res=None):
        """avk_compare_packed_submit: the batch is queued (its copies run beside the kernels of the batch submitted before) -> a Ticket; Ticket.wait() -> ResultBatch.
        `pbatch` and `res` must live in pinned memory (pinned_packed / pinned_results) for the copies to overlap anything; at most four tickets are in flight."""
        config = config or CompareConfig(enable_sequences=False)
        res = res if res is not None else ResultBatch(pbatch, sequences=False, group_metrics=False)
        pb, cfg, ro = pbatch.c_struct(), config.c_struct(), res.c_struct()
        handle = C.c_void_p()
        self._check(self.lib.avk_compare_packed_submit(self.handle, C.byref(pb), C.byref(cfg), C.byref(ro), C.byref(handle)))
        return Ticket(self, handle, res, (pbatch, pb, ro))

    def solve_compact(self, cbatch, config=None, res=None):
        """avk_compare_compact: solve_compare_region for every region of a batch in the compact form -> ResultBatch (indexed like the compact arrays)"""
        config = config or CompareConfig(enable_sequences=False)
        res = res if res is not None else ResultBatch(cbatch, sequences=False, group_metrics=False)
        cb, cfg, ro = cbatch.c_struct(), config.c_struct(), res.c_struct()
        self._check(self.lib.avk_compare_compact(self.handle, C.byref(cb), C.byref(cfg), C.byref(ro)))
        return res

    def pinned_results(self, batch, group_metrics=False, bp_groups=False, packed=False):
        """a ResultBatch whose arrays live in pinned memory (packed: as ResultBatch)"""
        res = ResultBatch(batch, sequences=False, group_metrics=group_metrics, bp_groups=bp_groups, packed=packed)
        for f in ("status", "ed_h1", "ed_h2", "n_optima", "type_present", "var_expected", "var_observed", "var_class", "var_zyg", "region_packed", "var_packed",
                  "group_metrics", "bp_off", "bp_groups", "bp_packed", "bp_spilled"):
            a = getattr(res, f, None)
            if a is None:
                continue
            b = self.host_array(a.shape, a.dtype)
            b[...] = a
            setattr(res, f, b)
        return res

    def expand_results(self, res, batch):
        """the wide arrays of a ResultBatch that holds the packed form (avk_results_expand)"""
        return res.expanded(self.lib, batch)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AardvarkAmdError("libaardvark_amd error %d: %s" % (rc, self.lib.avk_last_error(self.handle).decode()))

    def set_option(self, name, value):
        self._check(self.lib.avk_ctx_set_option(self.handle, name.encode(), int(value)))

    def set_stream(self, hip_stream):
        self._check(self.lib.avk_ctx_set_stream(self.handle, C.c_void_p(hip_stream)))

    def upload_reference(self, contigs):
        """contigs: list of bytes / uint8 arrays (ReferenceGenome::from_fasta + get_full_chromosome,
        reference src/main.rs:94, src/waffle_solver.rs:131)."""
        arrs = [np.frombuffer(c, dtype=np.uint8) if isinstance(c, (bytes, bytearray)) else np.ascontiguousarray(c, dtype=np.uint8)
                for c in contigs]
        ptrs = (u8p * len(arrs))(*[a.ctypes.data_as(u8p) for a in arrs])
        lens = (C.c_uint64 * len(arrs))(*[a.size for a in arrs])
        self._check(self.lib.avk_ref_upload(self.handle, len(arrs), ptrs, lens))
        self._contigs = arrs

    def solve_compare_regions(self, batch, config=None, group_metrics=True, bp_groups=False, packed=False):
        """solve_compare_region for every region of `batch` -> ResultBatch."""
        config = config or CompareConfig()
        res = ResultBatch(batch, sequences=bool(config.enable_sequences), group_metrics=group_metrics, bp_groups=bp_groups, packed=packed)
        cb, cfg, ro = batch.c_struct(), config.c_struct(), res.c_struct()
        self._check(self.lib.avk_compare_batch(self.handle, C.byref(cb), C.byref(cfg), C.byref(ro)))
        return res

    # --- resident form (benchmarks, pipelines that keep batches in HBM)
    def upload(self, batch):
        return ResidentBatch(self, batch)

    def compare_resident(self, rb, config=None, tally_dev_ptr=None):
        config = config or CompareConfig(enable_sequences=False)
        cfg = config.c_struct()
        self._check(self.lib.avk_compare_resident(self.handle, rb.handle, C.byref(cfg), C.c_void_p(tally_dev_ptr or 0)))

    def download(self, rb, sequences=False, group_metrics=True, packed=False):
        res = ResultBatch(rb.batch, sequences=sequences, group_metrics=group_metrics, packed=packed)
        ro = res.c_struct()
        self._check(self.lib.avk_results_download(self.handle, rb.handle, C.byref(ro)))
        return res

    def label_tallies(self, rb, n_labels, label_off, label_idx, out=None):
        """avk_label_tallies: per-label sums of the resident batch's per-region metric blocks (needs emit_group_metrics during
        compare_resident); returns / adds to a [n_labels, TALLY_LEN] uint64 array"""
        out = np.zeros((n_labels, TALLY_LEN), np.uint64) if out is None else out
        off = np.ascontiguousarray(label_off, np.uint64)
        idx = np.ascontiguousarray(label_idx, np.uint32)
        if idx.size == 0:
            idx = np.zeros(1, np.uint32)
        self.lib.avk_label_tallies.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        self._check(self.lib.avk_label_tallies(self.handle, rb.handle, n_labels, off.ctypes.data_as(C.POINTER(C.c_uint64)), idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def synchronize(self):
        self._check(self.lib.avk_synchronize(self.handle))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self.lib.avk_last_kernel_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def last_solver_ms(self):
        ms = C.c_float(0)
        self._check(self.lib.avk_last_solver_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def last_compare_was_one_shot(self):
        return bool(self.lib.avk_last_compare_was_one_shot(self.handle))

    def last_lane_ms(self):
        """HIP-event time from the start of the last step to the end of its lane-per-region launches (0 when it had none)"""
        ms = C.c_float(0)
        self._check(self.lib.avk_last_lane_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def last_lane_solved(self):
        """regions the lane-per-region kernel finished in the step whose results were downloaded last"""
        out = C.c_uint64(0)
        self._check(self.lib.avk_last_lane_solved(self.handle, C.byref(out)))
        return int(out.value)

    def last_wide_solved(self):
        """regions the wave-cooperative kernel of the large searches on small windows (avk_wide.inl) finished in the step whose results were downloaded last"""
        out = C.c_uint64(0)
        self._check(self.lib.avk_last_wide_solved(self.handle, C.byref(out)))
        return int(out.value)

    def last_tier_counts(self):
        out = (C.c_uint64 * 5)()
        self._check(self.lib.avk_last_tier_counts(self.handle, out))
        return [int(x) for x in out]

    def work_order(self, rb, want_order=True):
        """avk_debug_work_order: (order[n] or None, plan) of a device-packed resident batch; plan = {"class_c", "class_c_not_wide", "class_b", "lanes",
        "fast": [(first record, regions, head regions) per lane class]}"""
        import numpy as np
        n = rb.batch.n_regions
        order = np.zeros(max(n, 1), np.uint32) if want_order else None
        counts = (C.c_uint64 * 22)()
        self.lib.avk_debug_work_order.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        self._check(self.lib.avk_debug_work_order(self.handle, rb.handle, order.ctypes.data_as(C.POINTER(C.c_uint32)) if want_order else None, counts))
        c = [int(x) for x in counts]
        return (order[:n] if want_order else None), {"class_c": c[0], "class_c_not_wide": c[1], "class_b": c[2], "lanes": c[3],
                                                      "fast": [(c[4 + 3 * k], c[5 + 3 * k], c[6 + 3 * k]) for k in range(6)]}

    def debug_phase_cycles(self):
        out = (C.c_uint64 * 16)()
        self._check(self.lib.avk_debug_phase_cycles(self.handle, out))
        return [int(x) for x in out]

    def optimize_pairs(self, batch, max_branch_factor=50):
        """merge path: optimize_sequences(truth range, query range)[0].is_exact_match() per region
        (reference src/merge_solver.rs:135-147) -> (status int32[], is_exact uint8[])"""
        status = np.full(batch.n_regions, -1, np.int32)
        exact = np.zeros(max(batch.n_regions, 1), np.uint8)
        cb = batch.c_struct()
        self._check(self.lib.avk_optimize_pairs_batch(self.handle, C.byref(cb), max_branch_factor,
                                                      status.ctypes.data_as(C.POINTER(C.c_int32)), exact.ctypes.data_as(u8p)))
        return status, exact[:batch.n_regions]

    def algorithmic_bytes(self, batch, with_groups=True):
        cb = batch.c_struct()
        return int(self.lib.avk_algorithmic_bytes_ex(C.byref(cb), 1 if with_groups else 0))


def group_metrics_from_compact(batch, res):
    """avk_group_metrics_from_compact for every solved region: the full [n][13][22] blocks rebuilt on the host from the per-call outputs and the compact per-region
    BASEPAIR groups of `res` (regions with a non-zero status stay zero)"""
    lib = load_library()
    out = np.zeros((batch.n_regions, 13, 22), np.uint32)
    cb, ro = batch.c_struct(), res.c_struct()
    status = res.status if res.status is not None else (res.region_packed & np.uint64(0x7F))
    for r in range(batch.n_regions):
        if status[r] != 0:
            continue
        rc = lib.avk_group_metrics_from_compact(C.byref(cb), r, C.byref(ro), out[r].ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc != 0:
            raise AardvarkAmdError("avk_group_metrics_from_compact failed for region %d" % r)
    return out
