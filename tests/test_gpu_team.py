"""avk_region_kernel_team on the GPU: the hand-over of jobs between the four waves of a workgroup (LDS words, workgroup-scope fences around global memory).  Batches of
large windows (every region class C, the head of the class on the team launch), a genome slice (the few long windows beside lanes and the wide kernel), against the
oracle bit for bit, and against the same context with the team launch switched off."""
import os

import numpy as np
import pytest

import oracle_lib
from aardvark_amd import CompareConfig, synth

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)


def solve(contigs, batch, opts, gm=True):
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.upload_reference(contigs)
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=gm)
        return got, ctx.last_tier_counts()
    finally:
        ctx.close()


@pytest.mark.parametrize("head", [48, 4, 400])
def test_large_windows_with_the_head_of_the_class_on_teams(oracle, head):
    contig, bed, truth, query = synth.contig_calls(5, 3_000_000, 3_800 / 3_000_000, seed_ref=905, seed_query=906, str_frac=0.15, multi_frac=0.05)
    batch = synth.cluster_regions_v(contig, bed, truth, query, 1000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=True)
    got, tiers = solve([contig], batch, {"team_head_regions": head})
    assert got.diff(want) == []
    off, _ = solve([contig], batch, {"team_long_windows": 0})
    assert off.diff(want) == []


def test_every_region_on_a_team(oracle):
    """every region class C and not the wide kernel's, the whole class at the head: all of them through the team launch, many regions per workgroup"""
    from test_wide_parity import het_cluster_regions
    contigs, batch = het_cluster_regions(31, 300, n_sites=(2, 7), indel=0.3)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=True)
    got, _ = solve(contigs, batch, {"lane_kernel": 0, "wide_kernel": 0, "class_c_nodes_x2": 1, "solo_min_variants": 1, "lds_bytes_per_wave": 0, "team_head_regions": 1024})
    assert got.diff(want) == []


def test_genome_slice_with_its_long_windows_on_teams(oracle):
    contigs, batch = synth.config_genome(scale=0.05)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=False)
    for opts in ({}, {"lane_min_regions": 0, "lane_min_batch": 0}):
        got, _ = solve(contigs, batch, opts, gm=False)
        assert got.diff(want) == []


def test_team_regions_in_the_shared_big_slices(oracle):
    """per-wave slices too small for the head of the class: its regions are solved again in the shared big slices — another allocation, any distance from the sibling
    waves' own slices, where the genotype search dealt to a sibling keeps its nodes (hap_extend_seq once addressed everything by 32-bit offsets from the region's workspace)"""
    contig, bed, truth, query = synth.contig_calls(5, 3_000_000, 3_800 / 3_000_000, seed_ref=905, seed_query=906, str_frac=0.15, multi_frac=0.05)
    batch = synth.cluster_regions_v(contig, bed, truth, query, 1000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=True)
    got, tiers = solve([contig], batch, {"adaptive_ws": 0, "ws_bytes_per_wave": 1 << 18, "big_ws_bytes": 64 << 20, "team_head_regions": 64})
    assert got.diff(want) == []
    assert tiers[3] > 0, tiers  # (some regions did outgrow their slices)
