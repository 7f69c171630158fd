"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §4 / §8c).  The vectors in tests/golden/ are transcribed from the reference's
#[test] functions and doc-tests; each case names its file:line."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib
from aardvark_amd._abi import ALLELE, CLS, F, N_FIELDS, VT, ZYG, RegionBatch
from oracle_lib import b2p, u8p

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UMAX = 2**64 - 1


def gold(name):
    return json.load(open(os.path.join(GOLD, name)))


def P(b):
    arr, n = b2p(b)
    return C.cast(arr, u8p), n


# ---------------------------------------------------------------------------- DWFALite
class Dwfa:
    def __init__(self, lib, max_ed=UMAX, handle=None):
        self.lib = lib
        self.h = handle if handle is not None else lib.orc_dwfa_new(max_ed)

    def update(self, b, o):
        pb, nb = P(b)
        po, no = P(o)
        return self.lib.orc_dwfa_update(self.h, pb, nb, po, no)

    def finalize(self, b, o):
        pb, nb = P(b)
        po, no = P(o)
        return self.lib.orc_dwfa_finalize(self.h, pb, nb, po, no)

    @property
    def ed(self):
        return int(self.lib.orc_dwfa_ed(self.h))

    @property
    def wavefront(self):
        buf = (C.c_uint64 * 65536)()
        n = self.lib.orc_dwfa_wavefront(self.h, buf, 65536)
        return [int(buf[i]) for i in range(n)]

    def clone(self):
        return Dwfa(self.lib, handle=self.lib.orc_dwfa_clone(self.h))

    def __eq__(self, o):
        return bool(self.lib.orc_dwfa_equal(self.h, o.h))


def run_dwfa_case(lib, c):
    d = Dwfa(lib)
    mode = c["mode"]
    if mode == "script":
        for op, b, o, ed in c["steps"]:
            rc = d.update(b.encode(), o.encode()) if op == "update" else d.finalize(b.encode(), o.encode())
            assert rc == 0 and d.ed == ed, (c["name"], op)
        return
    b, o = c["baseline"].encode(), c["other"].encode()
    if mode == "finalize":
        assert d.finalize(b, o) == 0
    elif mode == "update_prefixes":
        for l in range(len(o)):
            assert d.update(b, o[:l + 1]) == 0
            if "ed_each" in c:
                assert d.ed == c["ed_each"]
    elif mode == "update_full":
        assert d.update(b, o) == 0
    if "ed_after_updates" in c:
        assert d.ed == c["ed_after_updates"], c["name"]
    if c.get("then_finalize"):
        assert d.finalize(b, o) == 0
    if "ed_final" in c:
        assert d.ed == c["ed_final"], c["name"]
    if "wavefront" in c:
        assert d.wavefront == c["wavefront"], c["name"]


@pytest.mark.parametrize("case", gold("dwfa.json")["cases"], ids=lambda c: c["name"])
def test_dwfa_known_answers(oracle, case):
    run_dwfa_case(oracle, case)


def test_dwfa_cloning(oracle):
    """dynamic_wfa.rs:425-450 test_cloning"""
    c = gold("dwfa.json")["cloning"]
    seq, alt = c["sequence"].encode(), c["alt_sequence"].encode()
    d, d2 = Dwfa(oracle), Dwfa(oracle)
    for l in range(len(alt)):
        d.update(seq, seq[:l + 1])
        d2.update(seq, alt[:l + 1])
        if seq[l] == alt[l]:
            assert d == d2
        else:
            assert not (d == d2)
            d2 = d.clone()
    assert d.ed == c["final_ed"] and d2.ed == c["final_ed"]


def test_dwfa_finalized_is_frozen(oracle):
    """AlreadyFinalized, dynamic_wfa.rs:69-71,184-186"""
    d = Dwfa(oracle)
    assert d.finalize(b"ACGT", b"ACGA") == 0
    assert d.update(b"ACGT", b"ACGA") == 2
    assert d.finalize(b"ACGT", b"ACGA") == 2


def test_dwfa_max_edit_distance_increments_first(oracle):
    """dynamic_wfa.rs:146-149: the distance is bumped before the limit check"""
    d = Dwfa(oracle, max_ed=0)
    assert d.update(b"ACGT", b"AGGT") == 1
    assert d.ed == 1 and d.wavefront == [1]


def test_dwfa_big_early_termination(oracle):
    """dynamic_wfa.rs:453-468"""
    c = gold("dwfa_big.json")
    b, o = c["baseline"].encode(), c["other"].encode()
    d = Dwfa(oracle)
    pb, nb = P(b)
    po, _ = P(o)
    for i in range(len(o)):
        assert oracle.orc_dwfa_update(d.h, pb, nb, po, i + 1) == 0
        assert d.ed <= c["max_ed_during_updates"]
    assert d.ed == c["ed_after_updates"]
    assert d.finalize(b, o) == 0
    assert d.ed == c["ed_after_finalize"]
    pa, na = P(b)
    pc, nc = P(o)
    assert oracle.orc_wfa_ed(pa, na, pc, nc) == c["ed_after_finalize"]


# ---------------------------------------------------------------------------- sequence_alignment
def test_edit_distance_and_wfa_ed(oracle):
    g = gold("sequence_alignment.json")
    for a, b, dist in g["cases"]:
        va, vb = bytes(g["vectors"][a]), bytes(g["vectors"][b])
        pa, na = P(va)
        pb, nb = P(vb)
        assert oracle.orc_edit_distance(pa, na, pb, nb) == dist, (a, b)
        assert oracle.orc_wfa_ed(pa, na, pb, nb) == dist, (a, b)


def test_wfa_equals_full_dp_random(oracle):
    """wfa_ed and the O(nm) DP are two routes to the same number (sequence_alignment.rs:3);
    the kernels use the wavefront form for the skipped-variant penalty (haplotype_dwfa.rs:199)."""
    rng = np.random.default_rng(7)
    for _ in range(400):
        n, m = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        alpha = int(rng.integers(1, 5))
        a = bytes(rng.integers(65, 65 + alpha, n, dtype=np.uint8))
        b = bytes(rng.integers(65, 65 + alpha, m, dtype=np.uint8))
        pa, na = P(a)
        pb, nb = P(b)
        assert oracle.orc_edit_distance(pa, na, pb, nb) == oracle.orc_wfa_ed(pa, na, pb, nb)
        assert oracle.orc_wfa_ed(pa, na, pb, nb) == oracle.orc_wfa_ed(pb, nb, pa, na)


# ---------------------------------------------------------------------------- node-level tests
class HapNode:
    def __init__(self, lib, n_haps, start, max_ed, ref):
        self.lib, self.ref = lib, ref
        self.h = lib.orc_hapnode_new(n_haps, start, max_ed)

    def extend(self, is_truth, variant, a1, a2, sync=-1):
        pos, a0, al1 = variant[0], variant[1].encode(), variant[2].encode()
        pr, nr = P(self.ref)
        p0, n0 = P(a0)
        p1, n1 = P(al1)
        ok = C.c_int(0)
        rc = self.lib.orc_hapnode_extend(self.h, pr, nr, 1 if is_truth else 0, pos, p0, n0, p1, n1, ALLELE[a1], ALLELE[a2], sync, C.byref(ok))
        assert rc == 0
        return bool(ok.value)

    def finalize(self, end):
        pr, nr = P(self.ref)
        assert self.lib.orc_hapnode_finalize(self.h, pr, nr, end) == 0

    def seq(self, hap, is_truth):
        buf = (C.c_uint8 * 4096)()
        n = self.lib.orc_hapnode_seq(self.h, hap, 1 if is_truth else 0, buf, 4096)
        return bytes(buf[:n]).decode()

    def alleles(self, hap, is_truth):
        buf = (C.c_uint8 * 4096)()
        n = self.lib.orc_hapnode_alleles(self.h, hap, 1 if is_truth else 0, buf, 4096)
        inv = {v: k for k, v in ALLELE.items()}
        return [inv[buf[i]] for i in range(n)]


@pytest.mark.parametrize("case", gold("haplotype_dwfa.json")["cases"], ids=lambda c: c["name"])
def test_haplotype_dwfa(oracle, case):
    ref = gold("haplotype_dwfa.json")["contig"].encode()
    n = HapNode(oracle, 1, case["region_start"], UMAX, ref)
    for s in case["steps"]:
        n.extend(s["is_truth"], s["variant"], s["allele"], s["allele"])
        if "ed_after" in s:
            assert oracle.orc_hapnode_ed(n.h, 0) == s["ed_after"]
        if "skip_after" in s:
            assert oracle.orc_hapnode_skip(n.h, 0) == s["skip_after"]
    n.finalize(case["region_end"])
    a = case["after_finalize"]
    assert oracle.orc_hapnode_ed(n.h, 0) == a["ed"]
    if "skip" in a:
        assert oracle.orc_hapnode_skip(n.h, 0) == a["skip"]
        assert oracle.orc_hapnode_cost(n.h) == a["cost"]
    if "truth_seq" in a:
        assert n.seq(0, True) == a["truth_seq"] and n.seq(0, False) == a["query_seq"]
        assert n.alleles(0, True) == a["truth_alleles"] and n.alleles(0, False) == a["query_alleles"]


def test_comparison_node(oracle):
    g = gold("query_optimizer.json")
    c = g["comparison_node"]
    n = HapNode(oracle, 2, c["region_start"], UMAX, g["contig"].encode())
    for s in c["steps"]:
        n.extend(s["is_truth"], s["variant"], s["a1"], s["a2"])
    n.finalize(c["region_end"])
    assert oracle.orc_hapnode_cost(n.h) == c["total_cost"]
    assert n.seq(0, False) == c["query_seq1"] and n.alleles(0, False) == c["query_alleles1"]
    assert n.seq(1, False) == c["query_seq2"] and n.alleles(1, False) == c["query_alleles2"]


def test_exact_match_node(oracle):
    g = gold("exact_gt_optimizer.json")
    c = g["exact_match_node"]
    n = HapNode(oracle, 1, c["region_start"], 0, g["contig"].encode())
    errors = 0
    for s in c["steps"]:
        n.extend(s["is_truth"], s["variant"], s["allele"], s["allele"])
        errors += 1 if s["is_error"] else 0
    n.finalize(c["region_end"])
    assert oracle.orc_hapnode_ed(n.h, 0) == c["edit_distance"]  # short-circuits at 1 (:516-517)
    assert errors == c["num_errors"]
    assert n.seq(0, True) == c["truth_seq"] and n.alleles(0, True) == c["truth_alleles"]
    assert n.seq(0, False) == c["query_seq"] and n.alleles(0, False) == c["query_alleles"]


# ---------------------------------------------------------------------------- optimizers
def one_region_batch(r, with_zyg=True):
    fix = lambda v: (v[0], v[1], v[2], v[3], v[4] if with_zyg and len(v) > 4 else "HomozygousAlternate")
    return RegionBatch.from_regions([{"start": r["start"], "end": r["end"],
                                      "truth": [fix(v) for v in r["truth"]], "query": [fix(v) for v in r["query"]]}])


@pytest.mark.parametrize("reg", gold("query_optimizer.json")["regions"], ids=lambda r: r["name"])
def test_optimize_sequences(oracle, reg):
    contig = gold("query_optimizer.json")["contig"].encode()
    b = one_region_batch(reg)
    T, Q = len(reg["truth"]), len(reg["query"])
    cap = 64
    ed = (C.c_uint64 * (2 * cap))()
    sk = (C.c_uint64 * (4 * cap))()
    tz = (C.c_uint8 * max(1, cap * T))()
    qz = (C.c_uint8 * max(1, cap * Q))()
    pr, nr = P(contig)
    cb = b.c_struct()
    n = oracle.orc_optimize_sequences(C.byref(cb), 0, pr, nr, 50, cap, ed, sk, tz, qz)
    assert n >= 1
    e = reg["expect"]
    assert (ed[0], ed[1]) == (e["ed1"], e["ed2"])
    for k, name in enumerate(["truth_vs1", "truth_vs2", "query_vs1", "query_vs2"]):
        if name in e:
            assert sk[k] == e[name]
    seqs = {}
    for s, name in enumerate(["truth_seq1", "truth_seq2", "query_seq1", "query_seq2"]):
        buf = (C.c_uint8 * 4096)()
        ln = oracle.orc_last_sequence(0, s, buf, 4096)
        seqs[name] = bytes(buf[:ln]).decode()
        if name in e:
            assert seqs[name] == e[name], name
    if "truth_zygosity" in e:
        assert [tz[i] for i in range(T)] == [ZYG[z] for z in e["truth_zygosity"]]
    assert [qz[i] for i in range(Q)] == [ZYG[z] for z in e["query_zygosity"]]


@pytest.mark.parametrize("reg", gold("exact_gt_optimizer.json")["regions"], ids=lambda r: r["name"])
def test_optimize_gt_alleles(oracle, reg):
    contig = gold("exact_gt_optimizer.json")["contig"].encode()
    b = one_region_batch(reg, with_zyg=False)
    T, Q = len(reg["truth"]), len(reg["query"])
    ta = (C.c_uint8 * max(T, 1))(*[ALLELE[a] for a in reg["truth_alleles"]])
    qa = (C.c_uint8 * max(Q, 1))(*[ALLELE[a] for a in reg["query_alleles"]])
    to = (C.c_uint8 * max(T, 1))()
    qo = (C.c_uint8 * max(Q, 1))()
    pr, nr = P(contig)
    cb = b.c_struct()
    n = oracle.orc_optimize_gt_alleles(C.byref(cb), 0, pr, nr, ta, qa, to, qo)
    e = reg["expect"]
    assert n == e["num_errors"]
    assert [to[i] for i in range(T)] == [ALLELE[a] for a in e["truth_alleles"]]
    assert [qo[i] for i in range(Q)] == [ALLELE[a] for a in e["query_alleles"]]


# ---------------------------------------------------------------------------- solve_compare_region
def check_region_expectations(res, r, reg, batch):
    e = reg["expect"]
    assert res.status[r] == 0, reg["name"]
    assert int(res.ed_h1[r]) + int(res.ed_h2[r]) == e["total_ed"], reg["name"]
    gm = res.group_metrics[r]
    joint = gm[0]
    assert list(joint[[F["GT_TRUTH_TP"], F["GT_TRUTH_FN"], F["GT_QUERY_TP"], F["GT_QUERY_FP"], F["GT_TRUTH_FN_GT"], F["GT_QUERY_FP_GT"]]]) == e["gt"]
    assert list(joint[F["HAP_TRUTH_TP"]:F["HAP_TRUTH_TP"] + 4]) == e["hap"]
    assert list(joint[F["BP_TRUTH_TP"]:F["BP_TRUTH_TP"] + 4]) == e["basepair"]
    for tname, vals in e.get("basepair_by_type", {}).items():
        assert list(gm[1 + VT[tname]][F["BP_TRUTH_TP"]:F["BP_TRUTH_TP"] + 4]) == vals
    to, tc = int(batch.t_off[r]), int(batch.t_cnt[r])
    qo, qc = int(batch.q_off[r]), int(batch.q_cnt[r])
    got_t = [[int(res.var_expected[to + i]), int(res.var_observed[to + i]), int(res.var_class[to + i])] for i in range(tc)]
    got_q = [[int(res.var_expected[qo + i]), int(res.var_observed[qo + i]), int(res.var_class[qo + i])] for i in range(qc)]
    assert got_t == [[x, o, CLS[c]] for x, o, c in e["truth_variant_data"]], reg["name"]
    assert got_q == [[x, o, CLS[c]] for x, o, c in e["query_variant_data"]], reg["name"]
    if res.sequences:
        assert [res.sequence(r, k).decode() for k in range(5)] == e["sequences"], reg["name"]
    # the 8 supported types always own an entry (waffle_solver.rs:384-445)
    for t in ["Snv", "Insertion", "Deletion", "Indel", "TrContraction", "TrExpansion", "SvDeletion", "SvInsertion"]:
        assert int(res.type_present[r]) & (1 << VT[t])


def solver_batch():
    g = gold("waffle_solver.json")
    regions = [{"start": r["start"], "end": r["end"], "truth": r["truth"], "query": r["query"]} for r in g["regions"]]
    return g, RegionBatch.from_regions(regions)


def test_solve_compare_region_known_answers(oracle):
    g, batch = solver_batch()
    res = oracle_lib.compare_batch(oracle, batch, [g["contig"].encode()], sequences=True)
    for r, reg in enumerate(g["regions"]):
        check_region_expectations(res, r, reg, batch)
    # tally = sum of the per-region blocks (writers/summary.rs:146-158)
    want = res.group_metrics.astype(np.uint64).sum(axis=0).reshape(-1)
    assert np.array_equal(res.tally[:want.size], want)
    assert res.tally[want.size] == len(g["regions"]) and res.tally[want.size + 1] == 0


def test_solve_compare_region_threads_agree(oracle):
    g, batch = solver_batch()
    a = oracle_lib.compare_batch(oracle, batch, [g["contig"].encode()], sequences=True, threads=1)
    b = oracle_lib.compare_batch(oracle, batch, [g["contig"].encode()], sequences=True, threads=4)
    assert a.diff(b) == []


def test_generate_haplotype_sequence(oracle):
    g = gold("waffle_solver.json")
    contig = g["contig"].encode()
    pr, nr = P(contig)
    for c in g["generate_haplotype_sequence"]:
        b = RegionBatch.from_regions([{"start": c["start"], "end": c["end"], "truth": c["variants"], "query": []}])
        cb = b.c_struct()
        z = (C.c_uint8 * len(c["variants"]))(*[ZYG[v[4]] for v in c["variants"]])
        for hap, key in ((0, "hap1"), (1, "hap2")):
            buf = (C.c_uint8 * 256)()
            fe = C.c_uint64(0)
            n = oracle.orc_generate_haplotype_sequence(C.byref(cb), 0, 0, pr, nr, z, hap, buf, 256, C.byref(fe))
            assert n >= 0
            assert [bytes(buf[:n]).decode(), fe.value] == c[key], c["name"]


def test_perform_basepair_compare(oracle):
    for c in gold("waffle_solver.json")["perform_basepair_compare"]:
        pr, nr = P(c["reference"].encode())
        pt, nt = P(c["truth"].encode())
        pq, nq = P(c["query"].encode())
        out = (C.c_uint64 * 4)()
        oracle.orc_basepair_compare(pr, nr, pt, nt, pq, nq, out)
        assert list(out) == c["expect"], c["ref"]


# ---------------------------------------------------------------------------- data types
def fields_vec(d):
    v = [0] * N_FIELDS
    for k, x in d.items():
        v[F[k]] = x
    return v


def test_group_metrics_rules(oracle):
    g = gold("data_types.json")
    for c in g["group_add_truth"]:
        buf = (C.c_uint64 * N_FIELDS)()
        assert oracle.orc_group_add_truth(buf, c["weight"], c["expected"], c["observed"]) == 0
        assert list(buf) == fields_vec(c["fields"]), c["ref"]
    for c in g["group_add_truth_errors"]:
        buf = (C.c_uint64 * N_FIELDS)()
        assert oracle.orc_group_add_truth(buf, c["weight"], c["expected"], c["observed"]) != 0
    for c in g["group_add_query"]:
        buf = (C.c_uint64 * N_FIELDS)()
        assert oracle.orc_group_add_query(buf, c["weight"], c["expected"], c["observed"]) == 0
        assert list(buf) == fields_vec(c["fields"]), c["ref"]
    for c in g["group_add_query_errors"]:
        buf = (C.c_uint64 * N_FIELDS)()
        assert oracle.orc_group_add_query(buf, c["weight"], c["expected"], c["observed"]) != 0
    for c in g["swap"]:
        other = (C.c_uint64 * N_FIELDS)()
        o = c["other_truth"]
        assert oracle.orc_group_add_truth(other, o["weight"], o["expected"], o["observed"]) == 0
        mine = (C.c_uint64 * N_FIELDS)()
        oracle.orc_group_swap(mine, other)
        assert list(mine) == fields_vec(c["fields"]), c["ref"]


def test_variant_metrics_rules(oracle):
    g = gold("data_types.json")
    for c in g["variant_metrics"]:
        out, tog = (C.c_uint8 * 3)(), (C.c_uint8 * 3)()
        assert oracle.orc_variant_metrics(c["expected"], c["observed"], out, tog) == 0
        assert list(out) == [c["expected"], c["observed"], CLS[c["class"]]]
        assert list(tog) == [c["toggled"][0], c["toggled"][1], CLS[c["toggled"][2]]]
    for c in g["variant_metrics_errors"]:
        out, tog = (C.c_uint8 * 3)(), (C.c_uint8 * 3)()
        assert oracle.orc_variant_metrics(c["expected"], c["observed"], out, tog) != 0
