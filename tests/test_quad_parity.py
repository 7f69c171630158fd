"""The quad solver for the expensive small regions (aardvark_amd/csrc/avk_quad.inl: four lanes per region — (child, haplotype) in the phasing
search, one genotype search per haplotype, (side, haplotype) in the metrics phase — on the rows and primitives of avk_lane.inl) against the
oracle, bit for bit, through the lane emulator, whose quad primitives are rendezvous of the quad's four fibers.  Every case also runs with
the option off (the same launches one lane per region): results must not depend on which kernel solved a region."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import synth

THREADS = 8


@pytest.fixture
def all_quads():
    """every lane launch 16 records wide, so that every region of a lane class goes through the quads"""
    lib = emu_lib.load()
    lib.emu_set_lane_width(16, 16)
    yield lib
    lib.emu_set_lane_width(64, 64)
    lib.emu_set_lane_width_three(16)
    lib.emu_set_lane_head_width(16)
    lib.emu_set_lane_node_cap(32)


def on_and_off(oracle, contigs, batch, min_quad_share=0.0, **kw):
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4, **{k: v for k, v in kw.items() if k == "max_branch_factor"})
    quad = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_quad=True, **kw)
    lane = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_quad=False, **kw)
    assert quad.diff(want) == []
    assert lane.diff(want) == []
    assert lane.quad_solved == 0
    assert quad.quad_solved >= min_quad_share * batch.n_regions
    return quad, want


def test_reference_known_answer_regions_on_quads(oracle, all_quads):
    contigs, batch = scenarios.golden()
    quad, _ = on_and_off(oracle, contigs, batch, n_waves=2)
    assert quad.quad_solved >= batch.n_regions - 1


@pytest.mark.parametrize("seed,kw", [(401, {}), (402, {"repeat_unit": b"CA"}), (403, {"repeat_unit": b"A", "max_len": 4}), (404, {"max_len": 16, "span": (20, 190)}),
                                     (405, {"repeat_unit": b"CAG", "related": 0.9}), (406, {"span": (4, 40), "max_len": 3})])
@pytest.mark.parametrize("max_vars", [2, 3])
def test_region_fuzz_on_quads(oracle, all_quads, seed, kw, max_vars):
    """SNVs, insertions, deletions, indels, overlapping and same-position calls, repeats; up to three calls per side"""
    contigs, batch = scenarios.fuzz_regions(seed, 500, max_vars=max_vars, **kw)
    on_and_off(oracle, contigs, batch, min_quad_share=0.1, n_waves=8)


def test_whole_genome_mix_on_quads(oracle, all_quads):
    contig, batch = synth.config_indel_mix_v2(n_truth=4000, contig_len=2_000_000)
    on_and_off(oracle, [contig], batch, min_quad_share=0.2, n_waves=16)


@pytest.mark.parametrize("quota", [1, 2, 3, 7])
def test_branch_quota_on_quads(oracle, all_quads, quota):
    """the per-depth quota (query_optimizer.rs:222-225) drops nodes in pop order: the quads pop in the reference's order"""
    for seed, mv in ((411, 2), (412, 3)):
        contigs, batch = scenarios.fuzz_regions(seed, 300, max_vars=mv, related=0.8)
        on_and_off(oracle, contigs, batch, min_quad_share=0.1, n_waves=8, max_branch_factor=quota)


@pytest.mark.parametrize("pool", [0, 1, 3, 8])
def test_kept_node_states_on_quads(oracle, all_quads, pool):
    """option lane_pool: a kept front is written by the lane that holds the haplotype and read by the two lanes that extend it"""
    for seed, kw in ((421, {"max_vars": 3, "repeat_unit": b"CA", "related": 0.7}), (422, {"max_vars": 2, "max_len": 12, "span": (30, 150)})):
        contigs, batch = scenarios.fuzz_regions(seed, 400, **kw)
        for quota in (50, 2):
            on_and_off(oracle, contigs, batch, min_quad_share=0.1, n_waves=8, lane_pool=pool, max_branch_factor=quota)


@pytest.mark.parametrize("width,cap", [(4, 8), (8, 250), (16, 16)])
def test_regions_per_wave_and_node_budget(oracle, all_quads, width, cap):
    """4, 8 or 16 quads of a wave at work (a short head is spread over more waves), node budgets of the three-call class: scheduling only"""
    all_quads.emu_set_lane_width(width, width)
    all_quads.emu_set_lane_width_three(width)
    all_quads.emu_set_lane_node_cap(cap)
    for seed, kw in ((431, {"max_vars": 3}), (432, {"max_vars": 3, "repeat_unit": b"CA", "related": 0.9})):
        contigs, batch = scenarios.fuzz_regions(seed, 500, **kw)
        on_and_off(oracle, contigs, batch, min_quad_share=0.1, n_waves=8)


def test_heads_only(oracle):
    """the default launch graph: the heads of the one- and two-call classes and the three-call class on quads, the rest 64 lanes wide"""
    contig, batch = synth.config_indel_mix_v2(n_truth=5000, contig_len=2_500_000)
    quad, _ = on_and_off(oracle, [contig], batch, n_waves=16)
    assert 0 < quad.quad_solved < quad.lane_solved


def test_outputs_without_the_blocks_and_with_compact_groups(oracle, all_quads):
    """tally-only outputs and the compact BASEPAIR groups (the i-th group of a region is written by lane i mod 4)"""
    contigs, batch = scenarios.fuzz_regions(441, 500, max_vars=3, related=0.7)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    got = emu_lib.compare_batch(batch, contigs, threads=THREADS, group_metrics=False, n_waves=8)
    assert got.quad_solved > 50
    assert np.array_equal(got.tally, want.tally)
    got = emu_lib.compare_batch(batch, contigs, threads=THREADS, group_metrics=False, bp_groups=True, n_waves=8)
    assert np.array_equal(got.tally, want.tally)
    full = emu_lib.compare_batch(batch, contigs, threads=THREADS, group_metrics=True, bp_groups=True, n_waves=8)
    assert full.diff(want) == []
    assert np.array_equal(got.bp_groups, full.bp_groups)


def test_merge_pairs_on_quads(oracle, all_quads):
    """optimize_sequences(..)[0].is_exact_match() per pair (merge_solver.rs:137-143): the search alone"""
    contigs, batch = scenarios.fuzz_regions(451, 400, max_vars=3, related=0.9)
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=4)
    st_e, ex_e = emu_lib.optimize_pairs(batch, contigs, threads=THREADS)
    assert np.array_equal(st_o, st_e) and np.array_equal(ex_o, ex_e)


def test_windows_with_other_symbols_are_handed_over(oracle, all_quads):
    contigs, batch = scenarios.fuzz_regions(461, 300, max_vars=2, contig_len=2500, alphabet=b"ACGT" * 50 + b"Nc")
    quad, _ = on_and_off(oracle, contigs, batch, n_waves=8)
    assert 0 < quad.quad_solved < batch.n_regions
