"""The asynchronous boundary (avk_compare_packed_submit / avk_wait: batches in flight inside one context, the copies of one under the kernels of another) on a
real MI355X: every batch's results are those of the synchronous call, bit for bit, whatever the order the tickets are waited for."""
import os

import numpy as np
import pytest

import oracle_lib
from aardvark_amd import CompactBatch, PackedBatch, ResultBatch, synth

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def job():
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    ctx.set_option("lane_min_regions", 0)
    ctx.set_option("lane_min_batch", 0)
    contigs, batch = synth.config_genome(scale=0.02)
    ctx.upload_reference(contigs)
    whole = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
    yield ctx, contigs, batch, whole
    ctx.close()


def same(a, b):
    return np.array_equal(a.region_packed, b.region_packed) and np.array_equal(a.var_packed, b.var_packed) and np.array_equal(a.tally, b.tally)


@pytest.mark.parametrize("order", [(0, 1, 2, 3), (3, 1, 0, 2), (2, 3, 1, 0)])
def test_tickets_in_any_order(job, oracle, order):
    ctx, contigs, batch, whole = job
    parts = [ctx.pinned_packed(p) for p in whole.split(4)]
    assert len(parts) == 4 and sum(p.n_regions for p in parts) == whole.n_regions
    want = [ctx.solve_packed(p, res=ResultBatch(p, sequences=False, group_metrics=False, packed="only")) for p in parts]  # the synchronous call
    tickets = [ctx.submit_packed(p, res=ctx.pinned_results(p, packed="only")) for p in parts]  # four in flight
    got = [None] * 4
    for k in order:
        got[k] = tickets[k].wait()
    for k in range(4):
        assert same(got[k], want[k]), k
    # ... and they are the oracle's: the job's tally over the four pieces, the per-call bytes of the whole batch
    ref = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=False)
    assert np.array_equal(sum(g.tally.astype(np.uint64) for g in got), ref.tally)
    whole_sync = ctx.solve_packed(whole, res=ResultBatch(whole, sequences=False, group_metrics=False, packed="only"))
    assert np.array_equal(np.concatenate([g.var_packed[:p.n_variants] for g, p in zip(got, parts)]), whole_sync.var_packed[:whole.n_variants])
    assert whole_sync.expanded(ctx.lib, batch).diff(ref) == []


def test_a_fifth_batch_must_wait_and_pageable_arrays_are_solved_at_once(job):
    import aardvark_amd
    ctx, contigs, batch, whole = job
    parts = whole.split(5)
    pinned = [ctx.pinned_packed(p) for p in parts]
    tickets = [ctx.submit_packed(p, res=ctx.pinned_results(p, packed="only")) for p in pinned[:4]]
    with pytest.raises(aardvark_amd.AardvarkAmdError):
        ctx.submit_packed(pinned[4], res=ctx.pinned_results(pinned[4], packed="only"))
    first = tickets[0].wait()
    fifth = ctx.submit_packed(pinned[4], res=ctx.pinned_results(pinned[4], packed="only"))  # a slot is free again
    rest = [t.wait() for t in tickets[1:]] + [fifth.wait()]
    sync = [ctx.solve_packed(p, res=ResultBatch(p, sequences=False, group_metrics=False, packed="only")) for p in parts]
    for g, w in zip([first] + rest, sync):
        assert same(g, w)
    # arrays the library cannot copy behind the caller's back: the ticket is complete when the submit returns
    t = ctx.submit_packed(parts[0], res=ResultBatch(parts[0], sequences=False, group_metrics=False, packed="only"))
    assert same(t.wait(), sync[0])
    with pytest.raises(RuntimeError):
        t.wait()


def test_wide_results_and_a_second_round_through_the_same_slots(job):
    ctx, contigs, batch, whole = job
    parts = [ctx.pinned_packed(p) for p in whole.split(3)]
    for _ in range(3):
        tickets = [ctx.submit_packed(p, res=ctx.pinned_results(p)) for p in parts]
        for t, p in zip(tickets, parts):
            got = t.wait()
            want = ctx.solve_packed(p, res=ResultBatch(p, sequences=False, group_metrics=False))
            assert got.diff(want) == []


def test_packed_basepair_groups_through_tickets(job):
    """bp_packed / bp_spilled / bp_groups of batches in flight against the synchronous call (the spill count is read when the ticket is waited for).  Where a region's
    groups land in the spill area is decided by an atomic counter, so the words of spilled regions differ from call to call: compared are the one-word regions, which
    regions spill, and the spilled groups as a multiset; the first region's groups are followed through its index."""
    ctx, contigs, batch, whole = job
    parts = [ctx.pinned_packed(p) for p in whole.split(3)]
    want = [ctx.solve_packed(p, res=ResultBatch(p, sequences=False, group_metrics=False, packed="only", bp_groups="packed")) for p in parts]
    tickets = [ctx.submit_packed(p, res=ctx.pinned_results(p, packed="only", bp_groups="packed")) for p in parts]
    for k in (2, 0, 1):
        got = tickets[k].wait()
        n = parts[k].n_regions
        g, w = got.bp_packed[:n], want[k].bp_packed[:n]
        spill_g, spill_w = (g >> np.uint32(31)) != 0, (w >> np.uint32(31)) != 0
        assert same(got, want[k]) and np.array_equal(spill_g, spill_w) and np.array_equal(g[~spill_g], w[~spill_w])
        m = int(got.bp_spilled[0])
        assert m == int(want[k].bp_spilled[0]) > 0
        rows_g, rows_w = got.bp_groups[:m], want[k].bp_groups[:m]
        assert np.array_equal(rows_g[np.lexsort(rows_g.T)], rows_w[np.lexsort(rows_w.T)])
        r = int(np.flatnonzero(spill_g)[0])  # (a region has at least two groups when it spills: the joint one and one per call type)
        ig, iw = int(g[r] & np.uint32(0x7FFFFFFF)), int(w[r] & np.uint32(0x7FFFFFFF))
        assert np.array_equal(rows_g[ig:ig + 2], rows_w[iw:iw + 2])
