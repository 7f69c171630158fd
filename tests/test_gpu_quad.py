"""The quad kernel (aardvark_amd/csrc/avk_quad.inl: four lanes per region in the narrow lane launches) on a real MI355X through the C-ABI,
against the oracle, bit for bit; every case also with the option off (lane_quad = 0: the same launches one lane per region)."""
import os

import numpy as np
import pytest

import oracle_lib
import scenarios
from aardvark_amd import synth

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)


def make_ctx(**opts):
    import aardvark_amd
    c = aardvark_amd.Context(0)
    c.set_option("lane_min_regions", 0)
    c.set_option("lane_min_batch", 0)
    for k, v in opts.items():
        c.set_option(k, v)
    return c


@pytest.fixture(scope="module")
def ctxs():
    """every lane launch 16 records wide: every region of a lane class goes through the quads (on) or through one lane of sixteen (off)"""
    wide16 = dict(lane_width_one=16, lane_width_two=16, lane_width_three=16)
    on, off = make_ctx(lane_quad=1, **wide16), make_ctx(lane_quad=0, **wide16)
    yield on, off
    on.close()
    off.close()


def on_and_off(ctxs, oracle, contigs, batch, max_branch_factor=50, group_metrics=True, min_lane_share=0.1):
    from aardvark_amd import CompareConfig
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, max_branch_factor=max_branch_factor, group_metrics=group_metrics)
    for c in ctxs:
        c.set_option("emit_group_metrics", 1 if group_metrics else 0)
        c.upload_reference(contigs)
        got = c.solve_compare_regions(batch, CompareConfig(enable_sequences=False, max_branch_factor=max_branch_factor), group_metrics=group_metrics)
        assert got.diff(want) == []
        assert c.last_lane_solved() >= min_lane_share * batch.n_regions
    return want


def test_reference_known_answer_regions_on_quads(ctxs, oracle):
    contigs, batch = scenarios.golden()
    on_and_off(ctxs, oracle, contigs, batch)


@pytest.mark.parametrize("seed,kw", [(401, {}), (402, {"repeat_unit": b"CA"}), (403, {"repeat_unit": b"A", "max_len": 4}), (404, {"max_len": 16, "span": (20, 190)}),
                                     (405, {"repeat_unit": b"CAG", "related": 0.9}), (406, {"span": (4, 40), "max_len": 3})])
@pytest.mark.parametrize("max_vars", [2, 3])
def test_region_fuzz_on_quads(ctxs, oracle, seed, kw, max_vars):
    contigs, batch = scenarios.fuzz_regions(seed, 6000, max_vars=max_vars, **kw)
    on_and_off(ctxs, oracle, contigs, batch)


@pytest.mark.parametrize("quota", [1, 2, 3, 7])
def test_branch_quota_on_quads(ctxs, oracle, quota):
    contigs, batch = scenarios.fuzz_regions(412, 4000, max_vars=3, related=0.8)
    on_and_off(ctxs, oracle, contigs, batch, max_branch_factor=quota)


def test_one_contig_full_density_on_quads(ctxs, oracle):
    """one contig of the benchmark workload at full density, all of its lane regions on quads; with and without the per-region blocks"""
    contig, batch = synth.config_indel_mix_v2(n_truth=int(synth.HG002_TRUTH_CALLS * synth.CHR20_LEN / sum(synth.GRCH38)), contig_len=synth.CHR20_LEN)
    for gm in (True, False):
        on_and_off(ctxs, oracle, [contig], batch, group_metrics=gm, min_lane_share=0.9)


@pytest.mark.parametrize("opts", [
    dict(lane_head_width=4, lane_width_three=4),
    dict(lane_head_width=8, lane_width_three=8, lane_node_cap=8),
    dict(lane_head_width=16, lane_node_cap=250),
])
def test_default_launch_graph_with_quads_of_any_width(oracle, opts):
    """the heads and the three-call class on 4, 8 or 16 quads per wave, the rest 64 lanes wide: scheduling only"""
    from aardvark_amd import CompareConfig
    ctx = make_ctx(**opts)
    try:
        contig, bed, truth, query = synth.contig_calls(0, 24_000_000, 70_000 / 24_000_000, seed_ref=81, seed_query=82, str_frac=0.15, multi_frac=0.05)
        batch = synth.cluster_regions_v(contig, bed, truth, query, 50)
        want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
        ctx.set_option("emit_group_metrics", 0)
        ctx.upload_reference([contig])
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
        assert got.diff(want) == []
        assert ctx.last_lane_solved() > 0.5 * batch.n_regions
    finally:
        ctx.close()


def test_merge_pairs_on_quads(ctxs, oracle):
    contigs, batch = scenarios.fuzz_regions(451, 6000, max_vars=3, related=0.9)
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=CPUS)
    for c in ctxs:
        c.upload_reference(contigs)
        st, ex = c.optimize_pairs(batch)
        assert np.array_equal(st_o, st) and np.array_equal(ex_o, ex)
