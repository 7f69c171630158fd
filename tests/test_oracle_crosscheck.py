"""oracle/oracle.cpp against a SECOND, independently written restatement of the reference solver (tests/golden/make_crosscheck.py: plain Python, written
from the Rust sources) on the behaviours no reference test pins: the per-depth quota of optimize_sequences hit hundreds of times, the 500-expansion
auto-fail pruning of optimize_gt_alleles, SV / TR / unsupported call types through the per-type BASEPAIR groups, overlapping calls.  The fixture
(tests/golden/crosscheck.json) holds that restatement's answers; the oracle must reproduce every field.  The emulator run at the end carries the check over to
the kernels' logic; the GPU parity suite compares the kernels with the oracle on the same scenario families."""
import json
import os

import numpy as np
import pytest

import emu_lib
import oracle_lib
from aardvark_amd import RegionBatch
from aardvark_amd._abi import VT, ZYG, CLS, F, N_FIELDS, VARIANT_TYPES

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crosscheck.json")))
ST = {"OK": 0, "BRANCH_FACTOR": 2, "NO_RESULTS": 3, "NO_GT_RESULT": 4, "UNKNOWN_ALLELE": 5, "BAD_ZYGOSITY": 6, "VARIANT_METRICS": 7, "TRUTH_FP": 8, "RECORD_BP": 9,
      "SEQ_MISMATCH": 10, "AUTOFAIL_OOB": 11}


def batch_of(sc):
    regs = []
    for r in sc["regions"]:
        conv = lambda vs: [(v[0], v[1].encode("latin1"), v[2].encode("latin1"), v[3], v[4], v[5]) for v in vs]
        regs.append({"start": r["start"], "end": r["end"], "truth": conv(r["truth"]), "query": conv(r["query"])})
    return RegionBatch.from_regions(regs)


def check(sc, got, batch, sequences=True):
    for r, exp in enumerate(sc["expect"]):
        where = "%s region %d" % (sc["name"], r)
        assert int(got.status[r]) == ST[exp["status"]], where
        if exp["status"] != "OK":
            continue
        assert (int(got.ed_h1[r]), int(got.ed_h2[r]), int(got.n_optima[r])) == (exp["ed1"], exp["ed2"], exp["n_optima"]), where
        for side, off, cnt in (("truth", int(batch.t_off[r]), int(batch.t_cnt[r])), ("query", int(batch.q_off[r]), int(batch.q_cnt[r]))):
            for k in range(cnt):
                ea, oa = exp[side][k]
                v = off + k
                assert (int(got.var_expected[v]), int(got.var_observed[v])) == (ea, oa), (where, side, k)
                if side == "truth":
                    cls = "TP" if ea == oa else "FN"
                else:  # toggled (compare_benchmark.rs:109-123): expected = what the query-as-truth scoring observed
                    cls = "TP" if ea == oa else "FP"
                assert int(got.var_class[v]) == CLS[cls], (where, side, k)
                assert int(got.var_zyg[v]) == ZYG[exp[side + "_zyg"][k]], (where, side, k)
        gm = got.group_metrics[r]
        present = 0
        for name, vals in exp["groups"].items():
            g = 0 if name == "joint" else 1 + VT[name]
            if name != "joint":
                present |= 1 << VT[name]
            assert gm[g].tolist() == vals, (where, name, gm[g].tolist(), vals)
        for t in range(len(VARIANT_TYPES)):
            if not (present >> t) & 1:
                assert not gm[1 + t].any(), (where, VARIANT_TYPES[t])
        assert int(got.type_present[r]) == present, where
        if sequences:
            for k in range(5):
                assert got.sequence(r, k) == exp["seqs"][k].encode("latin1"), (where, k)


@pytest.mark.parametrize("sc", FIX["scenarios"], ids=[s["name"] for s in FIX["scenarios"]])
def test_oracle_reproduces_the_independent_restatement(oracle, sc):
    batch = batch_of(sc)
    contig = [sc["contig"].encode("latin1")]
    got = oracle_lib.compare_batch(oracle, batch, contig, max_branch_factor=sc["max_branch_factor"], sequences=True, threads=4)
    check(sc, got, batch)


def test_fixture_reaches_the_unpinned_behaviours(oracle):
    """the scenarios do hit the quota, the auto-fail threshold and carry SV / TR types (so the test above says something about them)"""
    names = {s["name"]: s for s in FIX["scenarios"]}
    sc = names["autofail_homopolymer"]
    oracle_lib.compare_batch(oracle, batch_of(sc), [sc["contig"].encode("latin1")], threads=1)
    assert oracle_lib.stats(oracle)["max_pops_b"] > 500
    types = {v[3] for r in names["sv_tr_types"]["regions"] for v in r["truth"] + r["query"]}
    assert {"SvInsertion", "SvDeletion", "TrExpansion", "TrContraction", "SvDuplication", "Indel"} <= types
    assert sum(s["generator_stats"]["quota_drops"] for s in FIX["scenarios"] if s["name"].startswith("quota")) > 800
    assert max(e["n_optima"] for e in names["het_clusters_mbf50"]["expect"]) >= 8  # tied optima: their order decides the winner
    assert names["het_clusters_mbf3"]["generator_stats"]["quota_drops"] > 100
    assert names["autofail_homopolymer"]["generator_stats"]["autofail_prunings"] >= 5
    small, large = names["quota_repeats_mbf1"]["expect"], names["quota_repeats_mbf50"]["expect"]
    assert sum(a["ed1"] + a["ed2"] != b["ed1"] + b["ed2"] or a["groups"] != b["groups"] for a, b in zip(small, large)) >= 3  # the quota changes answers


@pytest.mark.parametrize("lane_kernel", [True, False])
def test_kernel_logic_reproduces_the_independent_restatement(lane_kernel):
    for sc in FIX["scenarios"]:
        batch = batch_of(sc)
        got = emu_lib.compare_batch(batch, [sc["contig"].encode("latin1")], max_branch_factor=sc["max_branch_factor"], sequences=True, lane_kernel=lane_kernel, threads=8)
        check(sc, got, batch)


def test_wide_kernel_logic_reproduces_the_independent_restatement():
    """the wave-cooperative kernel (avk_wide.inl) on the same answers: every region planned as class C, no sequence output (it writes none)"""
    solved = 0
    for sc in FIX["scenarios"]:
        batch = batch_of(sc)
        got = emu_lib.compare_batch(batch, [sc["contig"].encode("latin1")], max_branch_factor=sc["max_branch_factor"], lane_kernel=False, class_c_all=True, wide_lds_bytes=40 * 1024, threads=8)
        check(sc, got, batch, sequences=False)
        solved += got.wide_solved
    assert solved >= 200  # (long alleles, more than eight calls on a side, SV types with 40-base alleles: the wave-per-region code's)


@pytest.mark.gpu
@pytest.mark.parametrize("lane_kernel", [1, 0])
def test_gpu_reproduces_the_independent_restatement(lane_kernel):
    """the kernels on the GPU, through the C-ABI, against the same independently generated answers (no oracle in between)"""
    import aardvark_amd
    from aardvark_amd import CompareConfig
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("lane_kernel", lane_kernel)
        ctx.set_option("lane_min_regions", 0)
        ctx.set_option("lane_min_batch", 0)
        for sc in FIX["scenarios"]:
            batch = batch_of(sc)
            ctx.upload_reference([sc["contig"].encode("latin1")])
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=True, max_branch_factor=sc["max_branch_factor"]))
            check(sc, got, batch)
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False, max_branch_factor=sc["max_branch_factor"]))
            check(sc, got, batch, sequences=False)
        if not lane_kernel:  # once more with every region planned as class C: the wave-cooperative kernel sees it first
            ctx.set_option("class_c_nodes_x2", 1000)
            ctx.set_option("wide_lds_bytes", 40 * 1024)
            solved = 0
            for sc in FIX["scenarios"]:
                batch = batch_of(sc)
                ctx.upload_reference([sc["contig"].encode("latin1")])
                got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False, max_branch_factor=sc["max_branch_factor"]))
                check(sc, got, batch, sequences=False)
                solved += ctx.last_wide_solved()
            assert solved >= 100
    finally:
        ctx.close()
