"""The lane-per-region solver for small regions (aardvark_amd/csrc/avk_lane.inl: one region per lane, 2-bit sequences, search nodes
replayed from one-word queue entries) against the oracle, bit for bit, through the lane emulator.  The same scenarios run on the real
kernel in test_gpu_parity.py.  Results must not depend on which kernel solved a region: every case is also run with the lane code
switched off."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import RegionBatch, synth

THREADS = 8


def both_ways(oracle, contigs, batch, min_lane_share=0.0, **kw):
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4, **{k: v for k, v in kw.items() if k == "max_branch_factor"})
    lane = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=True, **kw)
    wave = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=False, **kw)
    assert lane.diff(want) == []
    assert wave.diff(want) == []
    assert wave.lane_solved == 0
    assert lane.lane_solved >= min_lane_share * batch.n_regions
    return lane, want


def test_reference_known_answer_regions_on_lanes(oracle):
    contigs, batch = scenarios.golden()
    lane, _ = both_ways(oracle, contigs, batch, n_waves=2)
    assert lane.lane_solved >= batch.n_regions - 1  # (one of the eight has three calls on a side)


@pytest.mark.parametrize("seed,kw", [(101, {}), (102, {"repeat_unit": b"CA"}), (103, {"repeat_unit": b"A", "max_len": 4}), (104, {"max_len": 16, "span": (20, 190)}),
                                     (105, {"repeat_unit": b"CAG", "related": 0.9}), (106, {"span": (4, 40), "max_len": 3})])
def test_small_region_fuzz(oracle, seed, kw):
    """at most two calls per side: SNVs, insertions, deletions, indels, overlapping and same-position calls, repeats"""
    contigs, batch = scenarios.fuzz_regions(seed, 400, max_vars=2, **kw)
    both_ways(oracle, contigs, batch, min_lane_share=0.1, n_waves=8)


def test_whole_genome_mix_on_lanes(oracle):
    """the density and the features of the benchmark workload (multi-allelic sites, repeat-run indels written at shifted positions)"""
    contig, batch = synth.config_indel_mix_v2(n_truth=5000, contig_len=2_500_000)
    lane, want = both_ways(oracle, [contig], batch, min_lane_share=0.9, n_waves=16)
    # the shifted representations are there and are resolved as matches: one call per side at different positions, no error on either haplotype
    t1 = (batch.t_cnt == 1) & (batch.q_cnt == 1)
    shifted = t1 & (batch.var_pos[batch.t_off.astype(np.int64) * t1] != batch.var_pos[batch.q_off.astype(np.int64) * t1])
    exact = shifted & (want.ed_h1 == 0) & (want.ed_h2 == 0) & (batch.var_type[batch.t_off.astype(np.int64) * t1] != 0)
    assert exact.sum() >= 20


@pytest.mark.parametrize("quota", [1, 2, 3, 7])
def test_branch_quota_on_lanes(oracle, quota):
    """max_branch_factor below the number of orientations: the per-depth quota (query_optimizer.rs:222-225) drops nodes in pop order"""
    contigs, batch = scenarios.fuzz_regions(111, 300, max_vars=2, related=0.8)
    both_ways(oracle, contigs, batch, min_lane_share=0.1, n_waves=8, max_branch_factor=quota)


def test_windows_with_other_symbols_are_handed_over(oracle):
    """N, IUPAC and lower-case bytes in the window or in an ALT allele: not this kernel's class, solved by the wave-per-region code"""
    contigs, batch = scenarios.fuzz_regions(9, 300, max_vars=2, contig_len=2500, alphabet=b"ACGT" * 50 + b"Nc")  # about 1 % other symbols
    lane, _ = both_ways(oracle, contigs, batch, n_waves=8)
    assert 0 < lane.lane_solved < batch.n_regions


def test_capacities_are_class_limits_not_errors(oracle):
    """regions beyond the small classes (three calls on a side, long windows, large edit-distance bounds) beside regions inside them"""
    contigs, batch = scenarios.fuzz_regions(121, 300, max_vars=3, max_len=24, span=(30, 260))
    lane, _ = both_ways(oracle, contigs, batch, n_waves=8)
    assert 0 < lane.lane_solved < batch.n_regions


def test_merge_pairs_on_lanes(oracle):
    """optimize_sequences(..)[0].is_exact_match() per pair (merge_solver.rs:137-143)"""
    contigs, batch = scenarios.fuzz_regions(131, 400, max_vars=2, related=0.9)
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=4)
    st_e, ex_e = emu_lib.optimize_pairs(batch, contigs, threads=THREADS)
    assert np.array_equal(st_o, st_e) and np.array_equal(ex_o, ex_e)
    assert ex_o.sum() > 20


@pytest.mark.parametrize("one,two", [(32, 32), (16, 16), (64, 16)])
def test_narrow_tiles(oracle, one, two):
    """a wave that takes 32 or 16 records of a tile at a time (options lane_width_one / lane_width_two) gives what 64-wide tiles give"""
    lib = emu_lib.load()
    lib.emu_set_lane_width(one, two)
    try:
        for seed, kw in ((201, {}), (202, {"repeat_unit": b"CA", "related": 0.9}), (203, {"max_len": 16, "span": (20, 190)})):
            contigs, batch = scenarios.fuzz_regions(seed, 700, max_vars=2, **kw)
            want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
            got = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=True, n_waves=8)
            assert got.diff(want) == []
            assert got.lane_solved >= 0.1 * batch.n_regions
    finally:
        lib.emu_set_lane_width(64, 64)


@pytest.mark.parametrize("head,cap", [(0, 250), (4, 8), (32, 16), (16, 250)])
def test_head_launch_and_node_budget_do_not_change_results(oracle, head, cap):
    """the head launch of a class (regions with estimated edits, narrow tiles: option lane_head_width) and the node budget of the
    three-call class (lane_node_cap: larger searches are handed to an HBM-tier launch behind the class) are scheduling only"""
    lib = emu_lib.load()
    lib.emu_set_lane_head_width(head)
    lib.emu_set_lane_node_cap(cap)
    try:
        for seed, kw in ((301, {"max_vars": 3}), (302, {"max_vars": 3, "repeat_unit": b"CA", "related": 0.9}), (303, {"max_vars": 2, "max_len": 16, "span": (20, 190)})):
            contigs, batch = scenarios.fuzz_regions(seed, 900, **kw)
            want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
            got = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=True, n_waves=8)
            assert got.diff(want) == []
            assert got.lane_solved >= 0.1 * batch.n_regions
    finally:
        lib.emu_set_lane_head_width(16)
        lib.emu_set_lane_node_cap(32)


@pytest.mark.parametrize("pool", [0, 1, 3, 8])
def test_kept_node_states_do_not_change_a_result(oracle, pool):
    """option lane_pool: the fronts of queued nodes with a distance (and of nodes whose last step was stopped early) kept in `pool` slots per lane, taken up again
    when the node is popped; without a free slot, or with 0 slots, the node's path is replayed.  Regions with many costly nodes, with the branch quota deciding
    and without; 16 records per wave and 64"""
    lib = emu_lib.load()
    for seed, kw, quota in ((121, {"max_vars": 3, "related": 0.5}, 50), (122, {"max_vars": 2, "repeat_unit": b"CA", "max_len": 6}, 50), (123, {"max_vars": 3, "related": 0.9, "max_len": 12}, 2),
                            (124, {"max_vars": 2, "span": (8, 60), "max_len": 4}, 1)):
        contigs, batch = scenarios.fuzz_regions(seed, 300, **kw)
        want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4, max_branch_factor=quota)
        for width in (64, 16):
            lib.emu_set_lane_width(width, width)
            try:
                got = emu_lib.compare_batch(batch, contigs, threads=THREADS, n_waves=8, max_branch_factor=quota, lane_pool=pool)
            finally:
                lib.emu_set_lane_width(64, 64)
            assert got.diff(want) == [], (seed, width)
            assert got.lane_solved > 0.2 * batch.n_regions
