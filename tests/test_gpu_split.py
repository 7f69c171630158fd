"""One large avk_compare_packed call as several batches in flight (context option split_parts, compare_packed_split of avk_host.hip): the regions of a batch are
independent (the reference maps over them, src/main.rs:251-268), so a range of regions of a packed batch is a packed batch of its own; the parts go through the
staging slots of the asynchronous boundary — part k + 1 is copied and packed while part k is solved — and the caller gets the arrays of the whole call, byte for byte:
records, per-call decisions, the tally, and the BASEPAIR groups (one spill list for the call)."""
import os

import numpy as np
import pytest

import oracle_lib
from aardvark_amd import CompactBatch, PackedBatch, ResultBatch, synth
from aardvark_amd.api import group_metrics_from_compact

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def job(oracle):
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    contigs, batch = synth.config_genome(scale=0.03)
    ctx.upload_reference(contigs)
    whole = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
    ref = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=True)
    yield ctx, contigs, batch, whole, ref
    ctx.close()


def solve(ctx, whole, parts, bp):
    ctx.set_option("split_parts", parts)
    res = ctx.pinned_results(whole, packed="only", bp_groups="packed" if bp else False)
    return ctx.solve_packed(whole, res=res)


@pytest.mark.parametrize("parts", [2, 3, 4])
def test_a_split_call_returns_the_arrays_of_the_whole_call(job, parts):
    ctx, contigs, batch, whole, ref = job
    one = solve(ctx, whole, 1, False)
    assert ctx.last_one_shot() if hasattr(ctx, "last_one_shot") else True
    got = solve(ctx, whole, parts, False)
    assert np.array_equal(got.region_packed, one.region_packed) and np.array_equal(got.var_packed, one.var_packed) and np.array_equal(got.tally, one.tally)
    assert got.expanded(ctx.lib, batch).diff(oracle_view(ref)) == []
    assert ctx.last_lane_solved() > 0.8 * batch.n_regions  # the statistics are the call's, not the last part's


def oracle_view(ref):
    return ref


@pytest.mark.parametrize("parts", [2, 4])
def test_a_split_call_returns_the_full_compare_benchmark(job, parts):
    """with the packed BASEPAIR groups: the parts spill into ONE list behind one counter; every region's whole 13 x 22 block rebuilt from the packed groups and the
    per-call decisions equals the oracle's (where a region's groups sit in the list depends on the order the device's lanes got there)"""
    ctx, contigs, batch, whole, ref = job
    one = solve(ctx, whole, 1, True)
    got = solve(ctx, whole, parts, True)
    assert np.array_equal(got.region_packed, one.region_packed) and np.array_equal(got.var_packed, one.var_packed) and np.array_equal(got.tally, one.tally)
    n = whole.n_regions
    g, w = got.bp_packed[:n], one.bp_packed[:n]
    spill_g, spill_w = (g >> np.uint32(31)) != 0, (w >> np.uint32(31)) != 0
    assert np.array_equal(spill_g, spill_w) and np.array_equal(g[~spill_g], w[~spill_w])
    m = int(got.bp_spilled[0])
    assert m == int(one.bp_spilled[0]) > 0
    rows_g, rows_w = got.bp_groups[:m], one.bp_groups[:m]
    assert np.array_equal(rows_g[np.lexsort(rows_g.T)], rows_w[np.lexsort(rows_w.T)])
    # spilled indices of every part point into the one list: all of them below the count, no two regions on the same rows
    idx = (g[spill_g] & np.uint32(0x7FFFFFFF)).astype(np.int64)
    assert idx.max() < m and np.unique(idx).size == idx.size
    n0 = min(n, 60_000)
    sub = batch.slice(0, n0)
    exp = got.expanded(ctx.lib, batch)
    exp.bp_packed, exp.bp_spilled, exp.bp_groups, exp.bp_off = got.bp_packed, got.bp_spilled, got.bp_groups, None
    full = group_metrics_from_compact(sub, exp)
    ok = ref.status[:n0] == 0
    assert np.array_equal(full[ok], ref.group_metrics[:n0][ok])
    # ... and from the LAST part as well (its words were written behind the other parts' spills)
    lo = n - min(n, 20_000)
    exp_tail = group_metrics_tail(batch, exp, lo)
    ok = ref.status[lo:] == 0
    assert np.array_equal(exp_tail[ok], ref.group_metrics[lo:][ok])


def group_metrics_tail(batch, exp, lo):
    import ctypes as C
    import aardvark_amd
    lib = aardvark_amd.load_library()
    out = np.zeros((batch.n_regions - lo, 13, 22), np.uint32)
    cb, ro = batch.c_struct(), exp.c_struct()
    for r in range(lo, batch.n_regions):
        if exp.status[r] != 0:
            continue
        assert lib.avk_group_metrics_from_compact(C.byref(cb), r, C.byref(ro), out[r - lo].ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    return out


def test_pageable_calls_are_not_split(job):
    ctx, contigs, batch, whole, ref = job
    ctx.set_option("split_parts", 1)  # the default: whole
    res = ctx.solve_packed(whole, res=ctx.pinned_results(whole, packed="only"))
    ctx.set_option("split_parts", 2)  # ... and pageable arrays are never split
    pageable = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
    res2 = ctx.solve_packed(pageable, res=ResultBatch(pageable, sequences=False, group_metrics=False, packed="only"))
    assert np.array_equal(res.region_packed, res2.region_packed) and np.array_equal(res.var_packed, res2.var_packed) and np.array_equal(res.tally, res2.tally)
