"""The wave-cooperative solver for large searches on small windows (aardvark_amd/csrc/avk_wide.inl: one region per wave, the sorted queue in
registers, up to 16 queue entries expanded per round on the 64 lanes, commits in the reference's order) against the oracle, bit for bit, through
the kernel-logic emulator.  The same scenarios run on the real kernel in test_gpu_wide.py.  Results must not depend on which kernel solved a
region: every case is also run with the wide code switched off."""
import json
import os

import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import RegionBatch, synth

THREADS = 8


def through_wide(oracle, contigs, batch, min_share=0.0, lane_kernel=False, **kw):
    """every region outside the lane classes is planned as class C, so that the wide code sees it first (40 KB of LDS per wave unless the case says otherwise:
    the stress regions here keep more nodes alive than a genome's, the default 16 KB hand more of them over)"""
    kw.setdefault("wide_lds_bytes", 40 * 1024)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4, **{k: v for k, v in kw.items() if k == "max_branch_factor"})
    wide = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=lane_kernel, class_c_all=True, **kw)
    off = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=lane_kernel, class_c_all=True, wide_kernel=False, **kw)
    assert wide.diff(want) == []
    assert off.diff(want) == []
    assert off.wide_solved == 0
    assert wide.wide_solved >= min_share * batch.n_regions, (wide.wide_solved, batch.n_regions)
    return wide, want


def het_cluster_regions(seed, n, n_sites=(3, 7), span=(90, 230), drop=0.1, shift=0.15, indel=0.0):
    """the regions that end a whole-genome step: a short window with several call pairs, most of them unphased heterozygous on both sides — a
    symmetric phasing search whose nodes all cost 0 until the end"""
    rng = np.random.default_rng(seed)
    contig = synth.ACGT[rng.integers(0, 4, size=6000, dtype=np.uint8)]
    zy = scenarios.ZY
    regions = []
    for _ in range(n):
        L = int(rng.integers(span[0], span[1]))
        start = int(rng.integers(0, contig.size - L))
        k = int(rng.integers(n_sites[0], n_sites[1] + 1))
        pos = np.sort(rng.choice(np.arange(start + 3, start + L - 12), size=k, replace=False))
        truth, query = [], []
        for p in pos:
            p = int(p)
            if rng.random() < indel:
                if rng.random() < 0.5:
                    ref, alt, vt = bytes(contig[p:p + 1]), bytes(contig[p:p + 1]) + bytes(synth.ACGT[rng.integers(0, 4, size=int(rng.integers(1, 6)), dtype=np.uint8)]), "Insertion"
                else:
                    rl = int(rng.integers(2, 7))
                    ref, alt, vt = bytes(contig[p:p + rl]), bytes(contig[p:p + 1]), "Deletion"
            else:
                ref, alt, vt = bytes(contig[p:p + 1]), bytes(synth._snv_alt(contig[p:p + 1], rng)), "Snv"
            zt = zy[0] if rng.random() < 0.8 else zy[int(rng.integers(0, 4))]
            u = rng.random()
            if u >= drop:
                truth.append((p, ref, alt, vt, zt))
            u = rng.random()
            if u >= drop:
                zq = zy[0] if rng.random() < 0.85 else zy[int(rng.integers(0, 4))]
                if rng.random() < shift and vt == "Snv":
                    alt = bytes(synth._snv_alt(contig[p:p + 1], rng))
                query.append((p, ref, alt, vt, zq))
        # overlapping calls would be rejected as unsorted only when positions decrease: equal positions are fine
        regions.append({"start": start, "end": start + L, "truth": truth, "query": query})
    return [bytes(contig)], RegionBatch.from_regions(regions)


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, {"n_sites": (4, 8), "drop": 0.0, "shift": 0.0}), (13, {"indel": 0.4}), (14, {"n_sites": (2, 5), "indel": 0.7, "drop": 0.3})])
def test_het_clusters(oracle, seed, kw):
    contigs, batch = het_cluster_regions(seed, 100, **kw)
    wide, want = through_wide(oracle, contigs, batch, min_share=0.6, n_waves=8)
    assert int(want.n_optima.max()) >= 4  # tied optima: the order the reference finds them in decides the winner


@pytest.mark.parametrize("seed,kw", [(21, {"max_vars": 3}), (22, {"max_vars": 5, "repeat_unit": b"CA", "related": 0.9}), (23, {"max_vars": 6, "repeat_unit": b"A", "max_len": 4}),
                                     (24, {"max_vars": 4, "max_len": 16, "span": (20, 200)}), (25, {"max_vars": 8, "repeat_unit": b"CAG", "related": 0.9}),
                                     (26, {"max_vars": 5, "span": (4, 40), "max_len": 3})])
def test_region_fuzz(oracle, seed, kw):
    """SNVs, insertions, deletions, indels, overlapping and same-position calls, repeats, up to ten calls on a side (beyond eight: handed over)"""
    contigs, batch = scenarios.fuzz_regions(seed, 180, **kw)
    through_wide(oracle, contigs, batch, min_share=0.3, n_waves=8)


@pytest.mark.parametrize("quota", [1, 2, 3, 7])
def test_branch_quota(oracle, quota):
    """max_branch_factor below the number of orientations: the per-depth quota (query_optimizer.rs:222-225) drops nodes in pop order — also nodes
    that were expanded ahead of their turn"""
    contigs, batch = het_cluster_regions(31, 120, n_sites=(3, 6))
    through_wide(oracle, contigs, batch, min_share=0.6, n_waves=8, max_branch_factor=quota)
    contigs, batch = scenarios.fuzz_regions(32, 200, max_vars=4, related=0.8)
    through_wide(oracle, contigs, batch, min_share=0.3, n_waves=8, max_branch_factor=quota)


def test_quota_regions_with_fourteen_query_hets_are_handed_over_or_solved(oracle):
    contigs, batch = scenarios.quota_regions(7)
    through_wide(oracle, contigs, batch, n_waves=4)
    through_wide(oracle, contigs, batch, n_waves=4, max_branch_factor=3)


def test_reference_known_answers_and_second_opinion_fixtures(oracle):
    contigs, batch = scenarios.golden()
    wide, _ = through_wide(oracle, contigs, batch, n_waves=2)
    assert wide.wide_solved >= 6
    contigs, batch, _ = scenarios.optimizer_golden_regions()
    through_wide(oracle, contigs, batch, n_waves=2)
    contigs, batch = scenarios.autofail_regions()  # more than 500 expansions in a genotype search: the wave-per-region code's
    through_wide(oracle, contigs, batch, n_waves=2)


def test_windows_with_other_symbols_and_long_alleles_are_handed_over(oracle):
    contigs, batch = scenarios.fuzz_regions(9, 200, max_vars=4, contig_len=2500, alphabet=b"ACGT" * 50 + b"Nc")
    wide, _ = through_wide(oracle, contigs, batch, n_waves=8)
    assert 0 < wide.wide_solved < batch.n_regions
    contigs, batch = scenarios.long_allele_regions()
    wide, _ = through_wide(oracle, contigs, batch.slice(2, 4), n_waves=2)  # the 600 bp SV pair and the TR pair (the 3 kbp pair takes the emulator a minute; it runs on the GPU)
    assert wide.wide_solved == 0


@pytest.mark.parametrize("seed,kw", [(81, {"max_vars": 2, "max_len": 30, "span": (60, 150)}), (82, {"max_vars": 3, "max_len": 24, "span": (40, 120), "related": 0.3}),
                                     (83, {"max_vars": 4, "max_len": 20, "span": (80, 160), "repeat_unit": b"CAG"})])
def test_large_edit_bounds_align_by_the_whole_wave(oracle, seed, kw):
    """edit bounds over 12: the alignments of the metrics phase are made one pair at a time by the whole wave (wfa_ed_wave), distances of tens"""
    contigs, batch = scenarios.fuzz_regions(seed, 300, **kw)
    wide, want = through_wide(oracle, contigs, batch, min_share=0.3, n_waves=8)
    assert int(np.maximum(want.ed_h1, want.ed_h2).max()) >= 20


@pytest.mark.parametrize("lds", [8 * 1024, 24 * 1024, 64 * 1024])
def test_lds_budget_is_a_class_limit(oracle, lds):
    """a region whose 2^T + 2^Q sequences do not fit the launch's LDS goes to the wave-per-region code"""
    contigs, batch = scenarios.fuzz_regions(41, 140, max_vars=8, related=0.9, span=(60, 230))
    through_wide(oracle, contigs, batch, min_share=0.1, n_waves=8, wide_lds_bytes=lds)


def test_outputs_without_group_blocks_and_with_compact_groups(oracle):
    contigs, batch = het_cluster_regions(51, 100, indel=0.3)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    got = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=False, class_c_all=True, group_metrics=False, bp_groups=True, wide_lds_bytes=40 * 1024)
    assert got.wide_solved >= 0.6 * batch.n_regions
    assert got.diff(want) == []
    off = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=False, class_c_all=True, wide_kernel=False, group_metrics=False, bp_groups=True)
    assert np.array_equal(got.bp_off, off.bp_off) and np.array_equal(got.bp_groups[:4 * int(got.bp_off[-1])], off.bp_groups[:4 * int(off.bp_off[-1])])


def test_hand_backs_of_the_three_call_lane_class(oracle):
    """with the lanes on, the three-call class hands its large searches (node budget) to the wide code"""
    lib = emu_lib.load()
    lib.emu_set_lane_node_cap(8)
    try:
        contigs, batch = het_cluster_regions(61, 300, n_sites=(2, 3), drop=0.0)
        want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
        got = emu_lib.compare_batch(batch, contigs, threads=THREADS, lane_kernel=True, n_waves=8)
        assert got.diff(want) == []
        assert got.wide_solved > 10 and got.lane_solved > 10
    finally:
        lib.emu_set_lane_node_cap(32)


def test_merge_pairs(oracle):
    """optimize_sequences(..)[0].is_exact_match() per pair (merge_solver.rs:137-143): pair batches have no class C, the wide code sees what the
    three-call lane class hands back"""
    lib = emu_lib.load()
    lib.emu_set_lane_node_cap(8)
    try:
        contigs, batch = het_cluster_regions(71, 300, n_sites=(2, 3), drop=0.02, shift=0.05)
        st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=4)
        lib.emu_set_lane_kernel(1)
        lib.emu_set_wide_kernel(1)
        st_e, ex_e = emu_lib.optimize_pairs(batch, contigs, threads=THREADS)
        assert int(lib.emu_last_wide_solved()) > 10
    finally:
        lib.emu_set_lane_node_cap(32)
    assert np.array_equal(st_o, st_e) and np.array_equal(ex_o, ex_e)
    assert 20 < ex_o.sum() < batch.n_regions



def test_merge_pairs_with_large_searches_are_planned_as_class_c(oracle):
    """context option pair_classes (default on since the end of round 4): pair batches plan their large searches as classes C and B, as compare batches do, so that the
    wide code takes them at the start of a step instead of the bulk launch's overflow at its end; merge_solver.rs:137-143 asks one bit per pair"""
    lib = emu_lib.load()
    contigs, batch = het_cluster_regions(72, 300, n_sites=(3, 7), drop=0.02, shift=0.05)
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=4)
    lib.emu_set_pair_classes.argtypes = [__import__("ctypes").c_int]
    seen = []
    for on in (1, 0):
        lib.emu_set_pair_classes(on)
        try:
            lib.emu_set_lane_kernel(1)
            lib.emu_set_wide_kernel(1)
            st_e, ex_e = emu_lib.optimize_pairs(batch, contigs, threads=THREADS)
            seen.append(int(lib.emu_last_wide_solved()))
        finally:
            lib.emu_set_pair_classes(1)
        assert np.array_equal(st_o, st_e) and np.array_equal(ex_o, ex_e)
    assert seen[0] > seen[1] and seen[0] > 30, seen
