"""A region on a team of wavefronts (avk_solver.inl, avk_region_kernel_team): the search posts its independent pieces as jobs — the haplotypes of a popped node's
children, made out of place; the alignments of the metrics — and the outputs are those of the one-wave search, bit for bit.  Here through the emulator, where the
owner wave takes every job itself (the decomposition); the hand-over between the waves of a workgroup runs on the GPU (tests/test_gpu_team.py)."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import synth


@pytest.fixture
def team():
    lib = emu_lib.load()
    lib.emu_set_team(1)
    yield lib
    lib.emu_set_team(0)


def through_the_hbm_tier(batch, contigs, **kw):
    """every region through the wave-per-region code of the HBM tier (no lanes, no wide kernel, no LDS tiers)"""
    return emu_lib.compare_batch(batch, contigs, lane_kernel=False, wide_kernel=False, lds_bytes=0, lds2_bytes=0, n_waves=4, **kw)


def test_reference_known_answers_on_a_team(oracle, team):
    contigs, batch = scenarios.golden()
    got = through_the_hbm_tier(batch, contigs, sequences=True)
    assert got.diff(oracle_lib.compare_batch(oracle, batch, contigs, sequences=True)) == []
    assert sum(got.tier_counts[2:4]) == batch.n_regions


@pytest.mark.parametrize("seed,kw", [(301, {}), (302, {"max_vars": 9, "max_len": 12}), (303, {"repeat_unit": b"CA", "max_vars": 4})])
def test_fuzz_regions_on_a_team(oracle, team, seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 60, **kw)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    assert through_the_hbm_tier(batch, contigs).diff(want) == []


def test_quota_autofail_and_odd_inputs_on_a_team(oracle, team):
    for sc in (scenarios.quota_regions(3), scenarios.autofail_regions(), scenarios.invalid_regions(), scenarios.non_acgt_regions(), scenarios.optimizer_golden_regions()):
        contigs, batch = sc[0], sc[1]
        want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
        assert through_the_hbm_tier(batch, contigs).diff(want) == []


def test_large_windows_on_a_team(oracle, team):
    """windows of kilobases with dozens of calls (--min-variant-gap 1000): the regions the team launch is for"""
    contig, bed, truth, query = synth.contig_calls(5, 160_000, 240 / 160_000, seed_ref=905, seed_query=906, str_frac=0.15, multi_frac=0.05)
    batch = synth.cluster_regions_v(contig, bed, truth, query, 1000)
    assert int((batch.t_cnt.astype(np.int64) + batch.q_cnt).max()) >= 10
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=4)
    got = through_the_hbm_tier(batch, [contig], ws_bytes=8 << 20)
    assert got.diff(want) == []


def test_merge_pairs_on_a_team(oracle, team):
    contigs, batch = scenarios.fuzz_regions(43, 60, max_vars=4, related=0.9)
    st, ex = emu_lib.optimize_pairs(batch, contigs)
    team.emu_set_team(0)
    st0, ex0 = emu_lib.optimize_pairs(batch, contigs)
    assert np.array_equal(st, st0) and np.array_equal(ex, ex0)
