"""Merge path: the pairwise exact-match primitive (optimize_sequences behind
avk_optimize_pairs_batch) and the host-side solve_merge_region classification, against the reference's
known-answer tests (src/merge_solver.rs:243-370) and, for the device kernels, against the oracle."""
import json
import os

import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import RegionBatch
from aardvark_amd.merge import MergeConfig, pair_batch, solve_merge_regions

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "merge_solver.json")))


def golden_regions():
    return [{"start": r["start"], "end": r["end"], "inputs": r["inputs"]} for r in GOLD["regions"]]


def classify(pairs_fn):
    got = []
    for r in GOLD["regions"]:
        res = solve_merge_regions(pairs_fn, [{"start": r["start"], "end": r["end"], "inputs": r["inputs"]}], MergeConfig(**r["config"]))
        got.append(res[0])
    return got


def expect():
    return [(0, tuple(r["expect"]) if len(r["expect"]) == 1 else (r["expect"][0], r["expect"][1])) for r in GOLD["regions"]]


def test_solve_merge_region_known_answers_oracle(oracle):
    contig = [GOLD["contig"].encode()]
    assert classify(lambda b, mbf: oracle_lib.optimize_pairs(oracle, b, contig, mbf)) == expect()


def test_solve_merge_region_known_answers_kernel_logic():
    contig = [GOLD["contig"].encode()]
    assert classify(lambda b, mbf: emu_lib.optimize_pairs(b, contig, mbf, threads=2)) == expect()


def test_variant_delta_length_rule(oracle):
    """merge_solver.rs:351-370: pairs whose net inserted length differs are not exact (and never reach the optimizer)"""
    g = GOLD["variant_delta_length"]
    contig = [b"ACGT" * 20]
    v = g["variants"]
    fixed = [(10, "A", "C", "Snv", v[0][4]), (12, "ACGTACGT", "A", "Deletion", v[1][4]), (25, "C", "CCC", "Insertion", v[2][4])]
    batch = RegionBatch.from_regions([{"start": 0, "end": 60, "truth": [fixed[1]], "query": [fixed[2]]},
                                      {"start": 0, "end": 60, "truth": [fixed[0]], "query": [fixed[0]]}])
    for fn in (lambda: oracle_lib.optimize_pairs(oracle, batch, contig), lambda: emu_lib.optimize_pairs(batch, contig, threads=2)):
        st, ex = fn()
        assert st.tolist() == [0, 0] and ex.tolist() == [0, 1]


def test_pairs_kernel_logic_matches_oracle_on_fuzz(oracle):
    for seed in (41, 42):
        contigs, batch = scenarios.fuzz_regions(seed, 150, related=0.85)
        st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=4)
        st_e, ex_e = emu_lib.optimize_pairs(batch, contigs, threads=8)
        assert np.array_equal(st_o, st_e) and np.array_equal(ex_o, ex_e)
        assert ex_o.any() and not ex_o.all()
    contigs, batch = scenarios.invalid_regions()
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs)
    st_e, ex_e = emu_lib.optimize_pairs(batch, contigs, threads=2)
    assert np.array_equal(st_o, st_e) and np.array_equal(ex_o, ex_e)


@pytest.mark.gpu
def test_pairs_on_gpu_match_oracle(oracle):
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    contig = [GOLD["contig"].encode()]
    ctx.upload_reference(contig)
    assert classify(lambda b, mbf: ctx.optimize_pairs(b, mbf)) == expect()
    for seed in (41, 42, 43):
        contigs, batch = scenarios.fuzz_regions(seed, 2000, related=0.85)
        ctx.upload_reference(contigs)
        st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=8)
        st_g, ex_g = ctx.optimize_pairs(batch)
        assert np.array_equal(st_o, st_g) and np.array_equal(ex_o, ex_g)
    contigs, batch = scenarios.invalid_regions()
    ctx.upload_reference(contigs)
    st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs)
    st_g, ex_g = ctx.optimize_pairs(batch)
    assert np.array_equal(st_o, st_g) and np.array_equal(ex_o, ex_g)
    # avk_merge_batch: the whole of solve_merge_region in one library call, per configuration of the known-answer tests
    from aardvark_amd.merge import merge_batch
    ctx.upload_reference(contig)
    for r, want in zip(GOLD["regions"], expect()):
        got = merge_batch(ctx, [{"start": r["start"], "end": r["end"], "inputs": r["inputs"]}], MergeConfig(**r["config"]))
        assert got[0] == want, r
    # and on three-input regions built from fuzzed call sets, against oracle pairs + the same classification
    contigs, batch = scenarios.fuzz_regions(44, 300, related=0.9)
    ctx.upload_reference(contigs)
    def variants(b, off, cnt):
        return [(int(b.var_pos[v]), bytes(b.allele_bytes[int(b.a0_off[v]):int(b.a0_off[v]) + int(b.a0_len[v])]),
                 bytes(b.allele_bytes[int(b.a1_off[v]):int(b.a1_off[v]) + int(b.a1_len[v])]), int(b.var_type[v]), int(b.var_zyg[v]),
                 int(b.var_raw_space[v])) for v in range(off, off + cnt)]
    multi = []
    for q in range(batch.n_regions):
        t = variants(batch, int(batch.t_off[q]), int(batch.t_cnt[q]))
        qv = variants(batch, int(batch.q_off[q]), int(batch.q_cnt[q]))
        multi.append({"start": int(batch.start[q]), "end": int(batch.end[q]), "inputs": [t, qv, t if q % 2 else qv]})
    for cfg in (MergeConfig(), MergeConfig(majority_voting_enabled=True), MergeConfig(no_conflict_enabled=True, conflict_selection=1)):
        want = solve_merge_regions(lambda b, mbf: oracle_lib.optimize_pairs(oracle, b, contigs, max_branch_factor=mbf, threads=8), multi, cfg)
        assert merge_batch(ctx, multi, cfg) == want
    ctx.close()


def test_device_merge_path_kernel_logic(oracle):
    """avk_merge_batch's device path (dp_expand_pairs -> pair solve -> dp_merge_classify, aardvark_amd/csrc/avk_devpack.inl) through the emulator: the reference's
    known-answer regions, and three-input regions from fuzzed call sets against oracle pairs + the library's host classification"""
    import ctypes as C
    from aardvark_amd.merge import AvkMergeConfig, AvkMultiBatch, MergeResult, MultiBatch
    from oracle_lib import ContigSet
    lib = emu_lib.load()
    lib.emu_merge_batch.argtypes = [C.POINTER(AvkMultiBatch), C.POINTER(oracle_lib.u8p), oracle_lib.u64p, C.c_uint32, C.POINTER(AvkMergeConfig), C.POINTER(C.c_int32), oracle_lib.u8p,
                                    oracle_lib.u64p, C.c_int]

    def device_merge(multi, contigs, config):
        mb = MultiBatch.from_regions(multi)
        cs = ContigSet(contigs)
        cfg = AvkMergeConfig(config.max_branch_factor, int(config.no_conflict_enabled), int(config.majority_voting_enabled), -1 if config.conflict_selection is None else int(config.conflict_selection))
        st, cl, mem = np.zeros(mb.n_regions, np.int32), np.zeros(mb.n_regions, np.uint8), np.zeros(mb.n_regions, np.uint64)
        cb = mb.c_struct()
        assert lib.emu_merge_batch(C.byref(cb), cs.ptrs, cs.lens, cs.n, C.byref(cfg), st.ctypes.data_as(C.POINTER(C.c_int32)), cl.ctypes.data_as(oracle_lib.u8p),
                                   mem.ctypes.data_as(oracle_lib.u64p), 4) == 0
        return MergeResult(st, cl, mem, mb.n_inputs).decoded()

    contig = [GOLD["contig"].encode()]
    for r, want in zip(GOLD["regions"], expect()):
        assert device_merge([{"start": r["start"], "end": r["end"], "inputs": r["inputs"]}], contig, MergeConfig(**r["config"]))[0] == want, r
    contigs, batch = scenarios.fuzz_regions(44, 70, related=0.9)

    def variants(b, off, cnt):
        return [(int(b.var_pos[v]), bytes(b.allele_bytes[int(b.a0_off[v]):int(b.a0_off[v]) + int(b.a0_len[v])]), bytes(b.allele_bytes[int(b.a1_off[v]):int(b.a1_off[v]) + int(b.a1_len[v])]),
                 int(b.var_type[v]), int(b.var_zyg[v]), int(b.var_raw_space[v])) for v in range(off, off + cnt)]
    multi = []
    for q in range(batch.n_regions):
        t = variants(batch, int(batch.t_off[q]), int(batch.t_cnt[q]))
        qv = variants(batch, int(batch.q_off[q]), int(batch.q_cnt[q]))
        multi.append({"start": int(batch.start[q]), "end": int(batch.end[q]), "inputs": [t, qv, t if q % 2 else qv, qv if q % 3 else []]})
    for cfg in (MergeConfig(), MergeConfig(majority_voting_enabled=True), MergeConfig(no_conflict_enabled=True, conflict_selection=1)):
        want = solve_merge_regions(lambda b, mbf: oracle_lib.optimize_pairs(oracle, b, contigs, max_branch_factor=mbf, threads=8), multi, cfg)
        assert device_merge(multi, contigs, cfg) == want


XMERGE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "merge_crosscheck.json")))


def xmerge_regions():
    conv = lambda vs: [(v[0], v[1].encode("latin1"), v[2].encode("latin1"), v[3], v[4], v[5]) for v in vs]
    return [{"start": r["start"], "end": r["end"], "inputs": [conv(i) for i in r["inputs"]]} for r in XMERGE["regions"]]


XNAMES = {"BasepairIdentical": "identical", "NoConflict": "no_conflict", "MajorityAgree": "majority", "ConflictSelection": "conflict_select", "Different": "different"}


def xmerge_expect(case):
    """the restatement names the reference's MergeClassification variants (merge_benchmark.rs:5-14); the library's decoded form uses its own short names"""
    return [(0, (XNAMES[e[0]],) if len(e) == 1 else (XNAMES[e[0]], e[1])) for e in case["expect"]]


def test_four_and_five_input_regions_against_the_independent_restatement(oracle):
    """solve_merge_region (merge_solver.rs:110-223) on regions of four and five call sets, four strategy settings: the answers of the second, independently written
    restatement (tests/golden/make_crosscheck.py) from the oracle's pairs + the library's classification, and from the device path's kernel logic"""
    contig = [XMERGE["contig"].encode("latin1")]
    multi = xmerge_regions()
    kinds = set()
    for case in XMERGE["cases"]:
        cfg = MergeConfig(**case["config"])
        want = xmerge_expect(case)
        kinds |= {w[1][0] for w in want}
        for n_inputs in (4, 5):  # (the regions of one call have one input count)
            idx = [i for i, r in enumerate(multi) if len(r["inputs"]) == n_inputs]
            sub, sub_want = [multi[i] for i in idx], [want[i] for i in idx]
            assert solve_merge_regions(lambda b, mbf: oracle_lib.optimize_pairs(oracle, b, contig, max_branch_factor=mbf, threads=4), sub, cfg) == sub_want
            assert solve_merge_regions(lambda b, mbf: emu_lib.optimize_pairs(b, contig, mbf, threads=8), sub, cfg) == sub_want
    assert {"identical", "no_conflict", "majority", "conflict_select", "different"} <= kinds


@pytest.mark.gpu
def test_four_and_five_input_regions_on_the_gpu():
    import aardvark_amd
    from aardvark_amd.merge import merge_batch
    ctx = aardvark_amd.Context(0)
    try:
        ctx.upload_reference([XMERGE["contig"].encode("latin1")])
        multi = xmerge_regions()
        for n_inputs in (4, 5):  # (a batch has one input count)
            idx = [i for i, r in enumerate(multi) if len(r["inputs"]) == n_inputs]
            for case in XMERGE["cases"]:
                want = xmerge_expect(case)
                assert merge_batch(ctx, [multi[i] for i in idx], MergeConfig(**case["config"])) == [want[i] for i in idx]
    finally:
        ctx.close()
