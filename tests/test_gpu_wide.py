"""The wave-cooperative kernel for large searches on small windows (aardvark_amd/csrc/avk_wide.inl) on a real MI355X through the C-ABI, against the
oracle, bit for bit; every case also with the kernel switched off (context option wide_kernel = 0)."""
import os

import numpy as np
import pytest

import oracle_lib
import scenarios
from test_wide_parity import het_cluster_regions

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def ctxs():
    import aardvark_amd
    on, off = aardvark_amd.Context(0), aardvark_amd.Context(0)
    for c in (on, off):
        c.set_option("lane_kernel", 0)           # nothing goes to the lanes ...
        c.set_option("class_c_nodes_x2", 1000)   # ... and every region is planned as class C: the wide kernel sees it first
        c.set_option("wide_lds_bytes", 40 * 1024)  # (the stress regions here keep more nodes alive than a genome's: the default 16 KB hand more of them over)
    off.set_option("wide_kernel", 0)
    yield on, off
    on.close()
    off.close()


def both_ways(ctxs, oracle, contigs, batch, min_share=0.0, max_branch_factor=50, group_metrics=True):
    from aardvark_amd import CompareConfig
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, max_branch_factor=max_branch_factor, group_metrics=group_metrics)
    out = []
    for c in ctxs:
        c.set_option("emit_group_metrics", 1 if group_metrics else 0)
        c.upload_reference(contigs)
        got = c.solve_compare_regions(batch, CompareConfig(enable_sequences=False, max_branch_factor=max_branch_factor), group_metrics=group_metrics)
        assert got.diff(want) == []
        out.append(c.last_wide_solved())
    assert out[1] == 0 and out[0] >= min_share * batch.n_regions, (out, batch.n_regions)
    return out[0], want


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, {"n_sites": (4, 8), "drop": 0.0, "shift": 0.0}), (13, {"indel": 0.4}), (14, {"n_sites": (2, 5), "indel": 0.7, "drop": 0.3})])
def test_het_clusters(ctxs, oracle, seed, kw):
    contigs, batch = het_cluster_regions(seed, 3000, **kw)
    n, want = both_ways(ctxs, oracle, contigs, batch, min_share=0.5)
    assert int(want.n_optima.max()) >= 4


@pytest.mark.parametrize("seed,kw", [(21, {"max_vars": 3}), (22, {"max_vars": 5, "repeat_unit": b"CA", "related": 0.9}), (23, {"max_vars": 6, "repeat_unit": b"A", "max_len": 4}),
                                     (24, {"max_vars": 4, "max_len": 16, "span": (20, 200)}), (25, {"max_vars": 8, "repeat_unit": b"CAG", "related": 0.9}),
                                     (26, {"max_vars": 5, "span": (4, 40), "max_len": 3})])
def test_region_fuzz(ctxs, oracle, seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 4000, **kw)
    both_ways(ctxs, oracle, contigs, batch, min_share=0.3)


@pytest.mark.parametrize("quota", [1, 2, 3, 7])
def test_branch_quota(ctxs, oracle, quota):
    contigs, batch = het_cluster_regions(31, 2000, n_sites=(3, 6))
    both_ways(ctxs, oracle, contigs, batch, min_share=0.5, max_branch_factor=quota)


def test_known_answers_other_symbols_long_alleles(ctxs, oracle):
    contigs, batch = scenarios.golden()
    both_ways(ctxs, oracle, contigs, batch)  # (eight regions: too few for a solo launch, the bulk takes them)
    contigs, batch = scenarios.fuzz_regions(9, 3000, max_vars=4, contig_len=2500, alphabet=b"ACGT" * 50 + b"Nc")
    n, _ = both_ways(ctxs, oracle, contigs, batch)
    assert 0 < n < batch.n_regions
    contigs, batch = scenarios.long_allele_regions()
    n, _ = both_ways(ctxs, oracle, contigs, batch)
    assert n == 0
    contigs, batch = scenarios.autofail_regions()
    both_ways(ctxs, oracle, contigs, batch)


def test_without_group_blocks(ctxs, oracle):
    contigs, batch = het_cluster_regions(51, 3000, indel=0.3)
    both_ways(ctxs, oracle, contigs, batch, min_share=0.5, group_metrics=False)


def test_hand_backs_of_the_three_call_lane_class_and_class_c_of_a_genome_slice():
    """defaults: a slice of the benchmark genome — class C and what the three-call lane class hands back go through the wide kernel"""
    import aardvark_amd
    from aardvark_amd import CompareConfig, synth
    orc = oracle_lib.load()
    contigs, batch = synth.config_genome(scale=0.06, threads=8)
    want = oracle_lib.compare_batch(orc, batch, contigs, threads=CPUS, group_metrics=False)
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("emit_group_metrics", 0)
        ctx.upload_reference(contigs)
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
        assert got.diff(want) == []
        assert ctx.last_wide_solved() > 100
    finally:
        ctx.close()


def test_small_slice_shares_do_not_overlap(oracle):
    """A workspace budget that leaves the class C launches eight workgroups' worth of HBM slices (as a batch with large adaptive slices does): the launch for the
    records that are not the wide kernel's takes its slices from the end of the solo launch's share and must leave that launch slices of its own — the two once met on
    a slice and a fuzz case came back wrong, now and then (profiles/r04_gpu_fuzz_classc.txt)"""
    import aardvark_amd
    from aardvark_amd import CompareConfig
    contigs, batch = scenarios.fuzz_regions(411003, 20000, max_vars=9, span=(40, 200))
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=False)
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("lane_min_regions", 0)
        ctx.set_option("class_c_nodes_x2", 1000)
        ctx.set_option("adaptive_ws", 0)
        ctx.set_option("ws_bytes_per_wave", 4 << 20)
        ctx.set_option("ws_budget_bytes", 1 << 30)  # 960 workgroups x 4 waves x 4 MB would be 15 GB: the shares shrink to eight workgroups
        ctx.upload_reference(contigs)
        for _ in range(4):
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
            assert got.diff(want) == []
            assert ctx.last_wide_solved() > 5000 and ctx.last_tier_counts()[2] > 1000
    finally:
        ctx.close()
