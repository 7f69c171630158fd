"""Known answers of the reference's optimizer and aligner tests on the KERNEL code paths (not only on the oracle):
the optimize_sequences regions of src/query_optimizer.rs:533-665 through the compare path — wave-per-region kernels with sequence output,
lane-per-region kernel without — and, on the GPU, alleles at the reference's 10 kbp limit and its 5,278-edit aligner vector as a region."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd._abi import ZYG


def check_optimizer_expectations(batch, expect, wave, lane):
    """wave: ResultBatch with sequences (wave-per-region code), lane: ResultBatch without (lane-per-region code)"""
    for i, e in enumerate(expect):
        for res in (wave, lane):
            assert res.status[i] == 0 and res.n_optima[i] == 1  # the expectations are on all_opt_haps[0]; one optimum: it is the winner
            assert (int(res.ed_h1[i]), int(res.ed_h2[i])) == (e["ed1"], e["ed2"])
            t0, q0 = int(batch.t_off[i]), int(batch.q_off[i])
            if "truth_zygosity" in e:
                assert res.var_zyg[t0:t0 + int(batch.t_cnt[i])].tolist() == [ZYG[z] for z in e["truth_zygosity"]]
            if "query_zygosity" in e:
                assert res.var_zyg[q0:q0 + int(batch.q_cnt[i])].tolist() == [ZYG[z] for z in e["query_zygosity"]]
        for k, name in ((1, "truth_seq1"), (2, "truth_seq2"), (3, "query_seq1"), (4, "query_seq2")):
            if name in e:
                assert wave.sequence(i, k) == e[name].encode(), (i, name)


def test_optimizer_known_answers_on_the_kernel_source(oracle):
    contigs, batch, expect = scenarios.optimizer_golden_regions()
    wave = emu_lib.compare_batch(batch, contigs, sequences=True, n_waves=2, threads=2)
    lane = emu_lib.compare_batch(batch, contigs, sequences=False, n_waves=2, threads=2)
    assert lane.lane_solved >= 3
    check_optimizer_expectations(batch, expect, wave, lane)
    assert wave.diff(oracle_lib.compare_batch(oracle, batch, contigs, sequences=True)) == []


@pytest.fixture(scope="module")
def ctx():
    import aardvark_amd
    c = aardvark_amd.Context(0)
    c.set_option("lane_min_regions", 0)
    yield c
    c.close()


@pytest.mark.gpu
def test_optimizer_known_answers_on_the_device(ctx):
    from aardvark_amd import CompareConfig
    contigs, batch, expect = scenarios.optimizer_golden_regions()
    ctx.upload_reference(contigs)
    wave = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=True))
    lane = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False))
    assert ctx.last_lane_solved() >= 3
    check_optimizer_expectations(batch, expect, wave, lane)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_alleles_at_the_size_limit_and_the_5278_edit_vector(ctx, oracle):
    """10,000-base alleles at the default max_branch_factor 50; the aligner vector of dynamic_wfa.rs:453-468 as a region: ed 5278 on both
    haplotypes; everything bit-identical to the oracle, sequences included; no region ends as a capacity failure"""
    from aardvark_amd import CompareConfig
    contigs, batch, big_ed = scenarios.max_allele_regions()
    want = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True, threads=3)
    ctx.upload_reference(contigs)
    got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=True, max_branch_factor=50))
    assert got.diff(want) == []
    assert got.status.tolist() == [0, 0, 0]
    assert (int(got.ed_h1[2]), int(got.ed_h2[2])) == (big_ed, big_ed)
    assert int(got.ed_h1[0]) + int(got.ed_h2[0]) >= 99  # the two 10 kbp insertions really differ in a hundred places
