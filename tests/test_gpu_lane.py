"""The lane-per-region kernel (aardvark_amd/csrc/avk_lane.inl) on a real MI355X through the C-ABI, against the oracle, bit for bit; every
case also with the kernel switched off (context option lane_kernel = 0: all regions through the wave-per-region kernels)."""
import os

import numpy as np
import pytest

import oracle_lib
import scenarios
from aardvark_amd import synth

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)


@pytest.fixture(scope="module")
def ctxs():
    import aardvark_amd
    on, off = aardvark_amd.Context(0), aardvark_amd.Context(0)
    on.set_option("lane_min_regions", 0)  # every class of the lane kernel, however small the batch
    off.set_option("lane_kernel", 0)
    yield on, off
    on.close()
    off.close()


def both_ways(ctxs, oracle, contigs, batch, min_lane_share=0.0, max_branch_factor=50, group_metrics=True):
    from aardvark_amd import CompareConfig
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, max_branch_factor=max_branch_factor, group_metrics=group_metrics)
    out = []
    for c in ctxs:
        c.set_option("emit_group_metrics", 1 if group_metrics else 0)
        c.upload_reference(contigs)
        got = c.solve_compare_regions(batch, CompareConfig(enable_sequences=False, max_branch_factor=max_branch_factor), group_metrics=group_metrics)
        assert got.diff(want) == []
        out.append(c.last_lane_solved())
    assert out[1] == 0 and out[0] >= min_lane_share * batch.n_regions
    return out[0], want


def test_reference_known_answer_regions_on_lanes(ctxs, oracle):
    contigs, batch = scenarios.golden()
    n, _ = both_ways(ctxs, oracle, contigs, batch)
    assert n >= batch.n_regions - 1


@pytest.mark.parametrize("seed,kw", [(101, {}), (102, {"repeat_unit": b"CA"}), (103, {"repeat_unit": b"A", "max_len": 4}), (104, {"max_len": 16, "span": (20, 190)}),
                                     (105, {"repeat_unit": b"CAG", "related": 0.9}), (106, {"span": (4, 40), "max_len": 3})])
def test_small_region_fuzz(ctxs, oracle, seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 6000, max_vars=2, **kw)
    both_ways(ctxs, oracle, contigs, batch, min_lane_share=0.1)


def test_whole_genome_mix_one_contig_full_density(ctxs, oracle):
    """one contig of the benchmark workload at full density (chr20-sized: about 75 k regions): per-variant decisions, per-region blocks,
    tally; multi-allelic sites and repeat-run indels at shifted positions included"""
    contig, batch = synth.config_indel_mix_v2(n_truth=int(synth.HG002_TRUTH_CALLS * synth.CHR20_LEN / sum(synth.GRCH38)), contig_len=synth.CHR20_LEN)
    n, want = both_ways(ctxs, oracle, [contig], batch, min_lane_share=0.9)
    assert int(want.tally[-2]) == batch.n_regions
    t1 = (batch.t_cnt == 1) & (batch.q_cnt == 1)
    shifted = t1 & (batch.var_pos[batch.t_off.astype(np.int64) * t1] != batch.var_pos[batch.q_off.astype(np.int64) * t1])
    assert (shifted & (want.ed_h1 == 0) & (want.ed_h2 == 0)).sum() > 100


@pytest.mark.parametrize("quota", [1, 2, 3, 7])
def test_branch_quota_on_lanes(ctxs, oracle, quota):
    contigs, batch = scenarios.fuzz_regions(111, 3000, max_vars=2, related=0.8)
    both_ways(ctxs, oracle, contigs, batch, min_lane_share=0.1, max_branch_factor=quota)


def test_windows_with_other_symbols_are_handed_over(ctxs, oracle):
    contigs, batch = scenarios.fuzz_regions(9, 3000, max_vars=2, contig_len=2500, alphabet=b"ACGT" * 50 + b"Nc")
    n, _ = both_ways(ctxs, oracle, contigs, batch)
    assert 0 < n < batch.n_regions


def test_capacities_are_class_limits_not_errors(ctxs, oracle):
    contigs, batch = scenarios.fuzz_regions(121, 3000, max_vars=3, max_len=24, span=(30, 260))
    n, _ = both_ways(ctxs, oracle, contigs, batch)
    assert 0 < n < batch.n_regions


def test_tally_only_outputs(ctxs, oracle):
    """what the benchmark runs: no per-region metric blocks"""
    contigs, batch = scenarios.fuzz_regions(141, 5000, max_vars=2, related=0.8)
    both_ways(ctxs, oracle, contigs, batch, min_lane_share=0.1, group_metrics=False)


def test_merge_pairs_on_lanes(ctxs, oracle):
    contigs, batch = scenarios.fuzz_regions(131, 4000, max_vars=2, related=0.9)
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=CPUS)
    for c in ctxs:
        c.upload_reference(contigs)
        st, ex = c.optimize_pairs(batch)
        assert np.array_equal(st_o, st) and np.array_equal(ex_o, ex)


def test_one_shot_path_equals_resident_path(ctxs, oracle):
    """avk_compare_batch (host arrays in, host arrays out; the batch is packed on the device, avk_devpack.inl) on a chr20-sized contig of the
    benchmark workload: every output equals the oracle's and the resident path's, with the lane kernel and without; windows with other symbols
    and regions beyond the lane classes are in the batch"""
    from aardvark_amd import CompareConfig
    contig, batch = synth.config_indel_mix_v2(n_truth=int(synth.HG002_TRUTH_CALLS * synth.CHR20_LEN / sum(synth.GRCH38)), contig_len=synth.CHR20_LEN)
    contig = contig.copy()
    rng = np.random.default_rng(5)
    contig[rng.integers(0, contig.size, size=contig.size // 5000)] = ord("N")  # some windows are not for the lanes
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
    on, off = ctxs
    for c in ctxs:
        c.set_option("emit_group_metrics", 0)
        c.upload_reference([contig])
    got = on.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
    assert on.last_compare_was_one_shot()
    assert got.diff(want) == []
    assert 0.8 * batch.n_regions < on.last_lane_solved() < batch.n_regions
    res = off.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
    assert off.last_compare_was_one_shot()
    assert off.last_lane_solved() == 0
    assert res.diff(want) == []
    rb = on.upload(batch)
    on.compare_resident(rb, CompareConfig(enable_sequences=False))
    assert on.download(rb, group_metrics=False).diff(want) == []
    rb.free()


@pytest.mark.parametrize("opts", [
    dict(lane_width_one=32, lane_width_two=16, lane_width_three=8, lane_head_width=4),
    dict(lane_head_width=0, lane_node_cap=8),
    dict(lane_head_width=32, lane_width_three=64),
    dict(lane_node_cap=250, lane_width_two=8),
    dict(lane_pairs=0, lane_quad=0),
    dict(lane_pairs=1, device_pack=0),
])
def test_scheduling_options_do_not_change_results(oracle, opts):
    """tile widths, head launches, node budget, edit estimates, stream layout: every combination gives the oracle's outputs, in the
    resident form (per-region blocks on) and through avk_compare_batch's one-shot path (blocks off)"""
    import aardvark_amd
    from aardvark_amd import CompareConfig
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("lane_min_regions", 0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        contig, bed, truth, query = synth.contig_calls(0, 24_000_000, 70_000 / 24_000_000, seed_ref=77, seed_query=78, str_frac=0.15, multi_frac=0.05)
        batch = synth.cluster_regions_v(contig, bed, truth, query, 50)
        for gm in (True, False):
            want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=gm)
            ctx.set_option("emit_group_metrics", 1 if gm else 0)
            ctx.upload_reference([contig])
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=gm)
            assert got.diff(want) == []
            assert ctx.last_lane_solved() > 0.5 * batch.n_regions
        assert ctx.last_compare_was_one_shot() or opts.get("device_pack", 1) == 0
    finally:
        ctx.close()


def test_merge_pairs_one_shot_path(ctxs, oracle):
    """avk_optimize_pairs_batch on a chr20-sized batch (packed on the device in the pair form, results unpacked on the device): status and
    exact-match flag of every pair equal the oracle's, with the lane kernel and without"""
    contig, batch = synth.config_indel_mix_v2(n_truth=int(synth.HG002_TRUTH_CALLS * synth.CHR20_LEN / sum(synth.GRCH38)), contig_len=synth.CHR20_LEN)
    contig = contig.copy()
    rng = np.random.default_rng(6)
    contig[rng.integers(0, contig.size, size=contig.size // 5000)] = ord("N")
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, [contig], threads=CPUS)
    on, off = ctxs
    for c in ctxs:
        c.upload_reference([contig])
        st, ex = c.optimize_pairs(batch)
        assert np.array_equal(st_o, st) and np.array_equal(ex_o, ex)
    on.upload_reference([contig])
    on.optimize_pairs(batch)
    assert on.last_compare_was_one_shot()
    assert on.last_lane_solved() > 0.8 * batch.n_regions
