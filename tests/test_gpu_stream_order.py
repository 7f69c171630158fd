"""The order the context's streams are made in (AVK_STREAM_ORDER, avk_ctx_create) places their hardware queues and is worth 10-40 % of a step
(profiles/r06_stream_order.txt) — and must never be worth a result: the same batch under the default, without placeholders, with the streams of the asynchronous calls
named, in reverse, and under strings that name nothing, through the resident step, the synchronous call and two calls in flight; every output against the oracle
(src/main.rs:251-268 is a map over independent regions: no order of launches may show)."""
import os

import numpy as np
import pytest

import oracle_lib
from aardvark_amd import CompactBatch, CompareConfig, PackedBatch, synth

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)
ORDERS = [None, "stwabcd", "stwxxabxxcdiopq", "dcbawts", "iopq", "xxxxxxxxxxxxxxxxxxxx", "", "z?"]


@pytest.fixture(scope="module")
def job(oracle):
    contigs, batch = synth.config_genome(scale=0.02)
    ref = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=False)
    return contigs, batch, ref


@pytest.mark.parametrize("order", ORDERS)
def test_results_do_not_depend_on_the_order_the_streams_are_made_in(job, order):
    import aardvark_amd
    contigs, batch, ref = job
    old = os.environ.pop("AVK_STREAM_ORDER", None)
    if order is not None:
        os.environ["AVK_STREAM_ORDER"] = order
    try:
        ctx = aardvark_amd.Context(0)
    finally:
        os.environ.pop("AVK_STREAM_ORDER", None)
        if old is not None:
            os.environ["AVK_STREAM_ORDER"] = old
    try:
        ctx.upload_reference(contigs)
        cfg = CompareConfig(enable_sequences=False)
        rb = ctx.upload(batch)
        for _ in range(3):  # queued steps: the launches of one step beside those of the next
            ctx.compare_resident(rb, cfg)
        got = ctx.download(rb, group_metrics=False)
        rb.free()
        whole = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
        sets = [ctx.pinned_results(whole, packed="only") for _ in range(2)]
        one = ctx.solve_packed(whole, res=sets[0])
        tickets = [ctx.submit_packed(whole, res=sets[k]) for k in range(2)]  # two in flight: the four streams of the asynchronous calls are made here
        for t in tickets:
            t.wait()
        for name in ("status", "ed_h1", "ed_h2", "n_optima", "var_expected", "var_observed", "var_class", "var_zyg"):
            assert np.array_equal(getattr(got, name), getattr(ref, name)), (order, name)
        assert np.array_equal(got.tally, ref.tally), order
        for k in range(2):
            assert np.array_equal(sets[k].region_packed, one.region_packed) and np.array_equal(sets[k].var_packed, one.var_packed), (order, k)
        assert one.expanded(ctx.lib, batch).diff(ref) == [], order
    finally:
        ctx.close()
