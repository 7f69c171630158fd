#!/usr/bin/env python3
"""A SECOND, independently written restatement of the reference's per-region solver — straight from the Rust sources, in plain Python, without looking at
oracle/oracle.cpp — used ONCE, in the build container, to generate tests/golden/crosscheck.json: known answers for the behaviours no reference test pins
(per-depth quota of optimize_sequences, the 500-expansion auto-fail of optimize_gt_alleles, SV / TR typed calls through the per-type BASEPAIR groups,
incompatible calls).  tests/test_oracle_crosscheck.py then requires oracle/oracle.cpp (and, through the parity suite, the kernels) to reproduce them.

What follows which reference lines:
  DWFALite                 src/dwfa/dynamic_wfa.rs:23-245           (update / extend / increase_edit_distance / finalize)
  HaplotypeTracker, HapDWFA src/dwfa/haplotype_dwfa.rs:17-245
  edit_distance            src/util/sequence_alignment.rs:20-51     (grid form, for the skip cost)
  wfa_ed                   src/util/sequence_alignment.rs:9-13
  optimize_sequences       src/query_optimizer.rs:166-365
  optimize_gt_alleles      src/exact_gt_optimizer.rs:108-357
  solve_compare_region     src/waffle_solver.rs:122-284 and its helpers :296-796
  metrics                  src/data_types/grouped_metrics.rs:32-280, summary_metrics.rs, variant_metrics.rs:25-110

usage: python tests/golden/make_crosscheck.py        (writes tests/golden/crosscheck.json)
"""
import heapq
import json
import os
import random

REF, ALT, UNK = "R", "A", "U"
TYPES = ["Snv", "Insertion", "Deletion", "Indel", "SvInsertion", "SvDeletion", "SvDuplication", "SvInversion", "SvBreakend", "TrContraction", "TrExpansion", "Unknown"]
SUPPORTED = ["Snv", "Insertion", "Deletion", "Indel", "TrContraction", "TrExpansion", "SvDeletion", "SvInsertion"]  # waffle_solver.rs:82-91
HETS = ("UnphasedHeterozygous", "PhasedHet01", "PhasedHet10")


class SolverError(Exception):
    pass


class MaxEditDistance(Exception):
    pass


# ---------------------------------------------------------------- dynamic_wfa.rs
class DWFA:
    def __init__(self, max_ed=None):
        self.ed = 0
        self.wf = [0]
        self.finalized = False
        self.max_ed = max_ed  # None = usize::MAX

    def clone(self):
        c = DWFA(self.max_ed)
        c.ed, c.wf, c.finalized = self.ed, list(self.wf), self.finalized
        return c

    def extend(self, base, other):  # :94-130
        wf, ed = self.wf, self.ed
        nb, no = len(base), len(other)
        for i in range(len(wf)):
            d = wf[i]
            while True:
                bo = d + ed - i
                if bo >= nb or d >= no or base[bo] != other[d]:
                    break
                d += 1
            wf[i] = d

    def increase(self, base, other):  # :140-173
        assert not self.finalized
        self.ed += 1
        if self.max_ed is not None and self.ed > self.max_ed:
            raise MaxEditDistance()
        new = [0] * (len(self.wf) + 2)
        for i, d in enumerate(self.wf):
            new[i] = max(new[i], d)
            new[i + 1] = max(new[i + 1], d + 1)
            new[i + 2] = max(new[i + 2], d + 1)
        self.wf = new
        self.extend(base, other)

    def max_base(self):  # :201-209
        return max(d + self.ed - i for i, d in enumerate(self.wf))

    def update(self, base, other):  # :68-84
        assert not self.finalized
        self.extend(base, other)
        while not (self.max_base() >= len(base)) and not (max(self.wf) >= len(other)):
            self.increase(base, other)

    def full_diagonal(self, base, other):  # :237-245
        return any(d + self.ed - i >= len(base) and d >= len(other) for i, d in enumerate(self.wf))

    def finalize(self, base, other):  # :183-198
        assert not self.finalized
        self.extend(base, other)
        while not self.full_diagonal(base, other):
            self.increase(base, other)
        self.finalized = True


def wfa_ed(a, b):
    d = DWFA()
    d.finalize(a, b)
    return d.ed


def grid_edit_distance(v1, v2):  # sequence_alignment.rs:20-51
    prev = list(range(len(v1) + 1))
    for i, c2 in enumerate(v2):
        row = [i + 1] + [0] * len(v1)
        for j, c1 in enumerate(v1):
            row[j + 1] = min(prev[j + 1] + 1, row[j] + 1, prev[j] + (0 if c1 == c2 else 1))
        prev = row
    return prev[len(v1)]


# ---------------------------------------------------------------- haplotype_dwfa.rs
class Tracker:
    def __init__(self, start):
        self.ref_pos = start
        self.alleles = []
        self.seq = bytearray()
        self.skip = 0

    def clone(self):
        c = Tracker(self.ref_pos)
        c.alleles, c.seq, c.skip = list(self.alleles), bytearray(self.seq), self.skip
        return c

    def copy_reference(self, reference, end):  # :236-245
        if self.ref_pos < end:
            self.seq += reference[self.ref_pos:end]
            self.ref_pos = end

    def extend_variant(self, reference, v, allele, ref_ext):  # :175-227
        vstart = v["pos"]
        self.copy_reference(reference, vstart)
        if allele == UNK:
            raise SolverError("UNKNOWN_ALLELE")
        ok = True
        if allele == ALT:
            if self.ref_pos <= vstart:
                self.seq += v["a1"]
                self.ref_pos = vstart + len(v["a0"])
            else:
                self.skip += grid_edit_distance(v["a0"], v["a1"])
                ok = False
        self.alleles.append(allele)
        if ref_ext is not None:
            self.copy_reference(reference, ref_ext)
        return ok


class HapDWFA:
    def __init__(self, start, max_ed):
        self.t = Tracker(start)
        self.q = Tracker(start)
        self.d = DWFA(max_ed)

    def clone(self):
        c = HapDWFA.__new__(HapDWFA)
        c.t, c.q, c.d = self.t.clone(), self.q.clone(), self.d.clone()
        return c

    def extend_variant(self, reference, is_truth, v, allele, sync):  # :46-66
        if is_truth:
            if sync is not None:
                self.q.copy_reference(reference, sync)
            ok = self.t.extend_variant(reference, v, allele, sync)
        else:
            if sync is not None:
                self.t.copy_reference(reference, sync)
            ok = self.q.extend_variant(reference, v, allele, sync)
        self.d.update(self.t.seq, self.q.seq)
        return ok

    def finalize(self, reference, end):  # :84-95
        self.t.copy_reference(reference, end)
        self.q.copy_reference(reference, end)
        self.d.update(self.t.seq, self.q.seq)
        self.d.finalize(self.t.seq, self.q.seq)

    def is_synchronized(self):  # :99-113
        return self.d.ed == 0 and len(self.t.seq) == len(self.q.seq) and self.t.ref_pos == self.q.ref_pos

    def set_alleles(self):
        return len(self.t.alleles) + len(self.q.alleles)

    def total_cost(self):
        return self.d.ed + self.t.skip + self.q.skip


def order_variants(tv, qv):  # query_optimizer.rs:372-381 (stable sort by position over truth.., query..)
    items = [(i, True) for i in range(len(tv))] + [(i, False) for i in range(len(qv))]
    items.sort(key=lambda it: tv[it[0]]["pos"] if it[1] else qv[it[0]]["pos"])
    return items


# ---------------------------------------------------------------- query_optimizer.rs:166-365
def optimize_sequences(reference, start, end, tv, tz, qv, qz, max_branch, stats=None):
    if max_branch <= 0:
        raise SolverError("BRANCH_FACTOR")
    order = order_variants(tv, qv)
    total = len(order)
    next_id = 0
    root = [next_id, HapDWFA(start, None), HapDWFA(start, None)]
    next_id += 1
    heap = [((0, root[0]), root)]
    best_ed = None
    best = []
    bucket = [0] * (total + 1)

    def cost(n):
        return n[1].total_cost() + n[2].total_cost()

    while heap:
        _, node = heapq.heappop(heap)  # lowest (cost, id) == highest (Reverse(cost), Reverse(id))
        if best_ed is not None and cost(node) > best_ed:
            continue
        oi = node[1].set_alleles()
        if bucket[oi] >= max_branch:
            if stats is not None:
                stats["quota_drops"] = stats.get("quota_drops", 0) + 1
            continue
        bucket[oi] += 1
        if oi == total:
            node[1].finalize(reference, end)
            node[2].finalize(reference, end)
            fc = cost(node)
            if best_ed is None or fc < best_ed:
                best_ed, best = fc, [node]
            elif fc == best_ed:
                best.append(node)
            continue
        vi, is_truth = order[oi]
        v, z = (tv[vi], tz[vi]) if is_truth else (qv[vi], qz[vi])
        if oi == total - 1:
            sync = end
        else:
            nvi, nt = order[oi + 1]
            sync = tv[nvi]["pos"] if nt else qv[nvi]["pos"]
        if z in HETS:
            if (not is_truth) or z == "UnphasedHeterozygous":
                for a1, a2 in ((REF, ALT), (ALT, REF)):
                    new = [next_id, node[1].clone(), node[2].clone()]
                    next_id += 1
                    new[1].extend_variant(reference, is_truth, v, a1, sync)
                    new[2].extend_variant(reference, is_truth, v, a2, sync)
                    heapq.heappush(heap, ((cost(new), new[0]), new))
            else:
                a1, a2 = (REF, ALT) if z == "PhasedHet01" else (ALT, REF)
                node[1].extend_variant(reference, is_truth, v, a1, sync)
                node[2].extend_variant(reference, is_truth, v, a2, sync)
                heapq.heappush(heap, ((cost(node), node[0]), node))
        else:
            if z != "HomozygousAlternate":
                raise SolverError("BAD_ZYGOSITY")  # assert_eq! panics, :315
            node[1].extend_variant(reference, is_truth, v, ALT, sync)
            node[2].extend_variant(reference, is_truth, v, ALT, sync)
            heapq.heappush(heap, ((cost(node), node[0]), node))
    if not best:
        raise SolverError("NO_RESULTS")

    def zyg(a1s, a2s):  # convert_alleles_to_zygosity :389-404
        out = []
        for a1, a2 in zip(a1s, a2s):
            if (a1, a2) == (REF, ALT):
                out.append("PhasedHet01")
            elif (a1, a2) == (ALT, REF):
                out.append("PhasedHet10")
            elif (a1, a2) == (ALT, ALT):
                out.append("HomozygousAlternate")
            else:
                raise SolverError("no impl")
        return out

    res = []
    for n in best:
        h1, h2 = n[1], n[2]
        res.append(dict(tz=zyg(h1.t.alleles, h2.t.alleles), qz=zyg(h1.q.alleles, h2.q.alleles), t1=bytes(h1.t.seq), t2=bytes(h2.t.seq), q1=bytes(h1.q.seq), q2=bytes(h2.q.seq),
                        ed1=h1.d.ed, ed2=h2.d.ed, tvs1=h1.t.skip, tvs2=h2.t.skip, qvs1=h1.q.skip, qvs2=h2.q.skip))
    return res


# ---------------------------------------------------------------- exact_gt_optimizer.rs:108-357
def optimize_gt_alleles(reference, start, end, tv, ta, qv, qa, stats=None):
    order = order_variants(tv, qv)
    total = len(order)
    next_id = 0
    # node = [id, hap, errors]
    root = [next_id, HapDWFA(start, 0), 0]
    next_id += 1

    def prio(n):  # (Reverse(errors), set - errors, Reverse(id)) as a min-heap key
        return (n[2], -(n[1].set_alleles() - n[2]), n[0])

    heap = [(prio(root), root)]
    best_err = None
    best = None
    min_sync = 0
    threshold = 500
    fail_index = 0
    fail_counts = 0
    expansions = 0

    def ext(node, is_truth, v, allele, sync, is_error):  # ExactMatchNode::extend_variant :396-414
        try:
            ok = node[1].extend_variant(reference, is_truth, v, allele, sync)
        except MaxEditDistance:
            assert node[1].d.ed != 0
            ok = False
        if is_error:
            node[2] += 1
        return ok

    while heap:
        _, node = heapq.heappop(heap)
        if best_err is not None and node[2] >= best_err:
            continue
        oi = node[1].set_alleles()
        if oi == total:
            try:
                node[1].finalize(reference, end)
            except MaxEditDistance:
                pass
            if node[1].d.ed == 0 and (best_err is None or node[2] < best_err):
                best_err, best = node[2], node
            continue
        if oi < min_sync:
            continue
        if node[1].is_synchronized():
            min_sync = oi
            fail_counts = 0
            fail_index = min_sync
        vi, is_truth = order[oi]
        v, a = (tv[vi], ta[vi]) if is_truth else (qv[vi], qa[vi])
        if oi == total - 1:
            sync = end
        else:
            nvi, nt = order[oi + 1]
            sync = tv[nvi]["pos"] if nt else qv[nvi]["pos"]
        if a == UNK:
            raise SolverError("UNKNOWN_ALLELE")
        if a == REF:
            ok = ext(node, is_truth, v, REF, sync, False)
            if ok and node[1].d.ed == 0:
                heapq.heappush(heap, (prio(node), node))
        else:
            for allele, is_error in ((REF, True), (ALT, False)):
                if oi < fail_index and allele != REF:
                    continue
                new = [next_id, node[1].clone(), node[2]]
                next_id += 1
                ok = ext(new, is_truth, v, allele, sync, is_error)
                if ok and new[1].d.ed == 0:
                    heapq.heappush(heap, (prio(new), new))
        expansions += 1
        fail_counts += 1
        if fail_counts >= threshold:
            if fail_index >= len(order):
                raise SolverError("AUTOFAIL_OOB")  # all_variant_order[auto_fail_index] panics
            sub, sub_truth = order[fail_index]
            kept = []
            for key, n in heap:
                alleles = n[1].t.alleles if sub_truth else n[1].q.alleles
                al = alleles[sub] if sub < len(alleles) else REF
                if al == REF:
                    kept.append((key, n))
            heapq.heapify(kept)
            heap = kept
            fail_index += 1
            fail_counts = 0
            if stats is not None:
                stats["autofail"] = stats.get("autofail", 0) + 1
    if stats is not None:
        stats["max_expansions"] = max(stats.get("max_expansions", 0), expansions)
    if best is None:
        raise SolverError("NO_GT_RESULT")
    return list(best[1].t.alleles), list(best[1].q.alleles), best[2]


# ---------------------------------------------------------------- metrics
def new_group():
    return dict(gt=[0, 0, 0, 0, 0, 0], hap=[0, 0, 0, 0], whap=[0, 0, 0, 0], bp=[0, 0, 0, 0], rbp=[0, 0, 0, 0])  # summary: truth_tp, truth_fn, query_tp, query_fp (+ fn_gt, fp_gt)


def group_add_truth(g, w, exp, obs):  # grouped_metrics.rs:183-227
    if exp == 0:
        raise SolverError("VARIANT_METRICS")  # ensure!(expected > 0)
    if exp < obs:
        raise SolverError("TRUTH_FP")
    if exp == obs:
        g["hap"][0] += exp
        g["whap"][0] += exp * w
        g["gt"][0] += 1
    else:
        g["hap"][0] += obs
        g["hap"][1] += exp - obs
        g["whap"][0] += obs * w
        g["whap"][1] += (exp - obs) * w
        g["gt"][1] += 1
        if obs > 0:
            g["gt"][4] += 1


def count(a):
    return 1 if a == ALT else 0


def decompose(z):  # phase_enums.rs decompose_alleles
    return {"Unknown": (UNK, UNK), "HomozygousReference": (REF, REF), "UnphasedHeterozygous": (REF, ALT), "PhasedHet01": (REF, ALT), "PhasedHet10": (ALT, REF),
            "HomozygousAlternate": (ALT, ALT)}[z]


def zyg_count(z):
    return {"Unknown": 0, "HomozygousReference": 0, "UnphasedHeterozygous": 1, "PhasedHet01": 1, "PhasedHet10": 1, "HomozygousAlternate": 2}[z]


def compare_expected_observed(variants, e1, o1, e2, o2):  # waffle_solver.rs:296-327 -> (joint group, per-type groups, per-variant (exp, obs))
    joint, by_type, per_var = new_group(), {}, []
    for v, a, b, c, d in zip(variants, e1, o1, e2, o2):
        exp, obs = count(a) + count(c), count(b) + count(d)
        if exp < obs:
            raise SolverError("TRUTH_FP")  # assert!(exp >= obs)
        w = wfa_ed(v["a0"], v["a1"])
        group_add_truth(joint, w, exp, obs)
        group_add_truth(by_type.setdefault(v["type"], new_group()), w, exp, obs)
        if exp == 0 and obs == 0:
            raise SolverError("VARIANT_METRICS")
        per_var.append((exp, obs))
    return joint, by_type, per_var


def generate_allele_sequence(reference, start, end, variants, alleles):  # waffle_solver.rs:726-778
    cur, seq, failed = start, bytearray(), 0
    for v, a in zip(variants, alleles):
        if a == REF:
            continue
        vpos = v["pos"]
        if vpos < cur:
            failed += wfa_ed(v["a0"], v["a1"])
            continue
        seq += reference[cur:vpos]
        cur = vpos
        if a == UNK:
            raise SolverError("UNKNOWN_ALLELE")
        seq += v["a1"]
        cur += len(v["a0"])
    seq += reference[cur:end]
    return bytes(seq), failed


def hap_sequence(reference, start, end, variants, zygs, hap):  # :683-712
    alleles = []
    for z in zygs:
        if z == "Unknown":
            raise SolverError("Unknown zygosity")
        a1, a2 = decompose(z)
        alleles.append(a1 if hap == 0 else a2)
    return generate_allele_sequence(reference, start, end, variants, alleles)


def basepair_compare(ref, truth, query):  # :622-649
    x, y, z = 2 * wfa_ed(ref, truth), 2 * wfa_ed(ref, query), 2 * wfa_ed(truth, query)
    tp = (x + y - z) // 2
    return [tp, x - tp, tp, y - tp]


def add4(dst, src):
    for i in range(4):
        dst[i] += src[i]


def solve_compare_region(reference, region, max_branch=50, stats=None):
    """-> dict(status=..) or the region's CompareBenchmark as plain data"""
    start, end = region["start"], region["end"]
    tv, qv = region["truth"], region["query"]
    tz, qz = [v["zyg"] for v in tv], [v["zyg"] for v in qv]
    optima = optimize_sequences(reference, start, end, tv, tz, qv, qz, max_branch, stats)
    results = []
    for oh in optima:
        th = [decompose(z) for z in oh["tz"]]
        qh = [decompose(z) for z in oh["qz"]]
        t1, t2 = [a for a, _ in th], [b for _, b in th]
        q1, q2 = [a for a, _ in qh], [b for _, b in qh]
        h1 = optimize_gt_alleles(reference, start, end, tv, t1, qv, q1, stats)
        h2 = optimize_gt_alleles(reference, start, end, tv, t2, qv, q2, stats)
        tstats = compare_expected_observed(tv, t1, h1[0], t2, h2[0])
        qstats = compare_expected_observed(qv, q1, h1[1], q2, h2[1])
        results.append((oh, h1, h2, tstats, qstats))
    # min_by_key keeps the FIRST minimum
    best = min(range(len(results)), key=lambda k: (results[k][1][2] + results[k][2][2], k))
    oh, h1, h2, (tj, tby, tpv), (qj, qby, qpv) = results[best]
    # add_swap_benchmark: query columns from the query-as-truth scores (grouped_metrics.rs:268-277, summary_metrics.rs set_query_from_truth)
    groups = {"joint": tj}
    for t, g in tby.items():
        groups[t] = g
    def swap(dst, src):
        dst["gt"][2], dst["gt"][3], dst["gt"][5] = src["gt"][0], src["gt"][1], src["gt"][4]
        dst["hap"][2], dst["hap"][3] = src["hap"][0], src["hap"][1]
        dst["whap"][2], dst["whap"][3] = src["whap"][0], src["whap"][1]
    swap(groups["joint"], qj)
    for t, g in qby.items():
        swap(groups.setdefault(t, new_group()), g)
    # add_basepair_stats :335-449
    ref_seq = reference[start:end]
    for hap in (0, 1):
        tseq, ted = hap_sequence(reference, start, end, tv, oh["tz"], hap)
        qseq, qed = hap_sequence(reference, start, end, qv, oh["qz"], hap)
        if tseq != (oh["t1"] if hap == 0 else oh["t2"]) or qseq != (oh["q1"] if hap == 0 else oh["q2"]):
            raise SolverError("SEQ_MISMATCH")
        add4(groups["joint"]["bp"], basepair_compare(ref_seq, tseq, qseq))
        add4(groups["joint"]["bp"], [0, 2 * ted, 0, 2 * qed])
        for ft in SUPPORTED:
            fq = [(v, z) for v, z in zip(qv, oh["qz"]) if v["type"] == ft]
            if fq:
                fseq, fed = hap_sequence(reference, start, end, [v for v, _ in fq], [z for _, z in fq], hap)
                m = basepair_compare(ref_seq, tseq, fseq)
                q_tp, q_fp = m[2], m[3] + 2 * fed
            else:
                q_tp, q_fp = 0, 0
            ftv = [(v, z) for v, z in zip(tv, oh["tz"]) if v["type"] == ft]
            if ftv:
                fseq, fed = hap_sequence(reference, start, end, [v for v, _ in ftv], [z for _, z in ftv], hap)
                m = basepair_compare(ref_seq, fseq, qseq)
                t_tp, t_fn = m[0], m[1] + 2 * fed
            else:
                t_tp, t_fn = 0, 0
            add4(groups.setdefault(ft, new_group())["bp"], [t_tp, t_fn, q_tp, q_fp])
    # add_record_basepair_stats :455-522 (totals from the INPUT zygosities)
    ttot, qtot, tby_t, qby_t = 0, 0, {}, {}
    for v in tv:
        c = zyg_count(v["zyg"]) * v["raw"]
        tby_t[v["type"]] = tby_t.get(v["type"], 0) + c
        ttot += c
    for v in qv:
        c = zyg_count(v["zyg"]) * v["raw"]
        qby_t[v["type"]] = qby_t.get(v["type"], 0) + c
        qtot += c
    jb = groups["joint"]["bp"]
    ttp, qtp = 2 * ttot - jb[1], 2 * qtot - jb[3]
    if ttp < jb[0] or qtp < jb[2]:
        raise SolverError("RECORD_BP")
    add4(groups["joint"]["rbp"], [ttp, jb[1], qtp, jb[3]])
    for t, g in groups.items():
        if t == "joint":
            continue
        b = g["bp"]
        add4(g["rbp"], [2 * tby_t.get(t, 0) - b[1], b[1], 2 * qby_t.get(t, 0) - b[3], b[3]])  # (u64 arithmetic: a negative value here would be a wrap in the reference)
    return dict(status="OK", ed1=oh["ed1"], ed2=oh["ed2"], n_optima=len(optima), truth=[list(p) for p in tpv], query=[[o, e] for e, o in qpv],  # query entries toggled (compare_benchmark.rs:109-123)
                truth_zyg=oh["tz"], query_zyg=oh["qz"], groups={t: [g["gt"][0], g["gt"][1], g["gt"][2], g["gt"][3], g["gt"][4], g["gt"][5]] + g["hap"] + g["whap"] + g["bp"] + g["rbp"]
                                                                 for t, g in groups.items()},
                seqs=[ref_seq.decode("latin1"), oh["t1"].decode("latin1"), oh["t2"].decode("latin1"), oh["q1"].decode("latin1"), oh["q2"].decode("latin1")])


# ---------------------------------------------------------------- scenarios
def V(pos, a0, a1, vtype, zyg, raw=None):
    a0 = a0 if isinstance(a0, bytes) else a0.encode()
    a1 = a1 if isinstance(a1, bytes) else a1.encode()
    return dict(pos=pos, a0=a0, a1=a1, type=vtype, zyg=zyg, raw=raw if raw is not None else max(len(a0), len(a1)))


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n)).encode()


def other_base(b, rng):
    return rng.choice([c for c in b"ACGT" if c != b])


def scenarios():
    rng = random.Random(20251002)
    out = []
    ZY = ["UnphasedHeterozygous", "PhasedHet01", "PhasedHet10", "HomozygousAlternate"]

    # 1. quota: many unphased query hets against a sparse truth, small branch factors
    contig = rand_seq(rng, 1400)
    regions = []
    for k in range(6):
        start = 40 + 220 * k
        end = start + 200
        pos = sorted(rng.sample(range(start + 5, end - 5), 9 + k % 3))
        query = [V(p, contig[p:p + 1], bytes([other_base(contig[p], rng)]), "Snv", "UnphasedHeterozygous") for p in pos]
        truth = [V(q["pos"], q["a0"], q["a1"], "Snv", rng.choice(ZY)) for q in query if rng.random() < 0.45]
        if k == 4:  # an insertion in the middle whose sequence can be spelled by a query call two bases later
            p = pos[3]
            truth.append(V(p, contig[p:p + 1], contig[p:p + 1] + b"GG", "Insertion", "PhasedHet01"))
            truth.sort(key=lambda v: v["pos"])
        regions.append(dict(start=start, end=end, truth=truth, query=query))
    for mb in (1, 2, 3, 7):
        out.append(dict(name="quota_mbf%d" % mb, contig=contig, regions=regions, max_branch_factor=mb))

    # 1b. quota in repeat runs: unphased insertions / deletions of whole units at shifted positions — the cheapest path at one depth is often a dead end later,
    # so a small branch factor CHANGES the answer (the same regions at 1, 2, 4 and 50)
    regions_rep, contigs_rep = [], []
    rr = random.Random(7)
    contig_rep = b""
    for _ in range(14):
        unit = rr.choice([b"CA", b"A", b"CAG", b"TTG"])
        base = len(contig_rep)
        piece = rand_seq(rr, 30) + unit * 12 + rand_seq(rr, 40)
        contig_rep += piece
        start, end, rs = base + 5, base + len(piece) - 5, base + 30

        def ins_at(k, n, z):
            p = rs + k * len(unit) - 1
            return V(p, contig_rep[p:p + 1], contig_rep[p:p + 1] + unit * n, "Insertion", z)

        def del_at(k, n, z):
            p = rs + k * len(unit) - 1
            return V(p, contig_rep[p:p + 1 + n * len(unit)], contig_rep[p:p + 1], "Deletion", z)
        truth = [rr.choice([ins_at, del_at])(rr.randrange(0, 8), rr.randrange(1, 3), rr.choice(ZY)) for _ in range(rr.randrange(1, 4))]
        query = [rr.choice([ins_at, del_at])(rr.randrange(0, 8), rr.randrange(1, 3), "UnphasedHeterozygous") for _ in range(rr.randrange(1, 5))]
        for _ in range(rr.randrange(0, 3)):
            p = rr.randrange(start + 2, end - 2)
            query.append(V(p, contig_rep[p:p + 1], bytes([other_base(contig_rep[p], rr)]), "Snv", "UnphasedHeterozygous"))
        truth.sort(key=lambda v: v["pos"])
        query.sort(key=lambda v: v["pos"])
        regions_rep.append(dict(start=start, end=end, truth=truth, query=query))
    for mb in (1, 2, 4, 50):
        out.append(dict(name="quota_repeats_mbf%d" % mb, contig=contig_rep, regions=regions_rep, max_branch_factor=mb))

    # 2. auto-fail: single-base deletions spread over a homopolymer; more of them on one side than on the other
    contig2 = b"CG" + b"A" * 120 + b"TC" + b"G" * 40
    regions = []
    for nt, nq, zy in ((8, 5, "HomozygousAlternate"), (9, 5, "UnphasedHeterozygous"), (5, 8, "PhasedHet01"), (10, 6, "HomozygousAlternate"), (10, 4, "UnphasedHeterozygous")):
        truth = [V(4 + 3 * i, "AA", "A", "Deletion", zy if zy != "UnphasedHeterozygous" else "PhasedHet10") for i in range(nt)]
        query = [V(50 + 3 * i, "AA", "A", "Deletion", zy) for i in range(nq)]
        regions.append(dict(start=0, end=len(contig2) - 30, truth=truth, query=query))
    out.append(dict(name="autofail_homopolymer", contig=contig2, regions=regions, max_branch_factor=50))

    # 3. SV / TR typed calls, mixed with small ones, incl. an unsupported type (SvDuplication) and incompatible (overlapping) calls
    contig3 = rand_seq(rng, 2600)
    c3 = contig3
    ins40 = rand_seq(rng, 40)
    ins41 = bytearray(ins40)
    ins41[17] = other_base(ins41[17], rng)
    rep = b"CAG" * 9
    contig3 = bytearray(contig3)
    contig3[1500:1500 + len(rep)] = rep
    contig3 = bytes(contig3)
    c3 = contig3
    regions = [
        # SV insertion vs nearly the same SV insertion, plus an SNV only truth has
        dict(start=100, end=400, truth=[V(200, c3[200:201], c3[200:201] + ins40, "SvInsertion", "HomozygousAlternate"), V(260, c3[260:261], bytes([other_base(c3[260], rng)]), "Snv", "PhasedHet10")],
             query=[V(200, c3[200:201], c3[200:201] + bytes(ins41), "SvInsertion", "UnphasedHeterozygous")]),
        # SV deletion in truth spelled as two deletions in the query; raw_allele_space larger than the alleles
        dict(start=500, end=820, truth=[V(600, c3[600:661], c3[600:601], "SvDeletion", "PhasedHet01", raw=80)],
             query=[V(600, c3[600:631], c3[600:601], "Deletion", "PhasedHet01"), V(631, c3[631:662], c3[631:632], "Deletion", "UnphasedHeterozygous")]),
        # TR expansion / contraction in a CAG run against plain insertion / deletion of the same units at shifted positions
        dict(start=1440, end=1600, truth=[V(1499, c3[1499:1500], c3[1499:1500] + b"CAGCAG", "TrExpansion", "HomozygousAlternate")],
             query=[V(1505, c3[1505:1506], c3[1505:1506] + b"CAGCAG", "Insertion", "PhasedHet10"), V(1511, c3[1511:1512], c3[1511:1512] + b"CAGCAG", "Insertion", "PhasedHet01")]),
        dict(start=1440, end=1600, truth=[V(1499, c3[1499:1506], c3[1499:1500], "TrContraction", "UnphasedHeterozygous"), V(1560, c3[1560:1561], bytes([other_base(c3[1560], rng)]), "Snv", "HomozygousAlternate")],
             query=[V(1508, c3[1508:1515], c3[1508:1509], "Deletion", "UnphasedHeterozygous"), V(1560, c3[1560:1561], bytes([other_base(c3[1560], rng)]), "Snv", "UnphasedHeterozygous")]),
        # an unsupported type (counted in GT / HAP groups, never in the per-type BASEPAIR filter) beside a supported one
        dict(start=1800, end=2050, truth=[V(1850, c3[1850:1851], c3[1850:1851] + c3[1851:1871], "SvDuplication", "PhasedHet10"), V(1900, c3[1900:1903], c3[1900:1901], "Deletion", "HomozygousAlternate")],
             query=[V(1850, c3[1850:1851], c3[1850:1851] + c3[1851:1871], "SvInsertion", "UnphasedHeterozygous"), V(1900, c3[1900:1903], c3[1900:1901], "Deletion", "PhasedHet01")]),
        # overlapping calls on one haplotype: the second is skipped and costs its edit distance
        dict(start=2100, end=2400, truth=[V(2200, c3[2200:2212], c3[2200:2201], "Deletion", "HomozygousAlternate"), V(2205, c3[2205:2206], bytes([other_base(c3[2205], rng)]), "Snv", "HomozygousAlternate"),
                                          V(2300, c3[2300:2302], b"TTTT" if c3[2300:2301] != b"T" else b"GGGG", "Indel", "PhasedHet01")],
             query=[V(2200, c3[2200:2212], c3[2200:2201], "SvDeletion", "UnphasedHeterozygous"), V(2300, c3[2300:2301], b"T" if c3[2300:2301] != b"T" else b"G", "Snv", "UnphasedHeterozygous")]),
    ]
    out.append(dict(name="sv_tr_types", contig=contig3, regions=regions, max_branch_factor=50))

    # 4. random small regions (all four zygosities, SNV / insertion / deletion / indel, related and unrelated sides)
    contig4 = rand_seq(rng, 3000)
    regions = []
    for k in range(40):
        L = rng.randrange(30, 120)
        start = rng.randrange(0, len(contig4) - L)
        end = start + L

        def rv():
            kind = rng.randrange(4)
            rl = 1 if kind in (0, 1) else rng.randrange(2, 6)
            pos = rng.randrange(start, end - rl + 1)
            ref = contig4[pos:pos + rl]
            if kind == 0:
                return V(pos, ref, bytes([other_base(ref[0], rng)]), "Snv", rng.choice(ZY))
            if kind == 1:
                return V(pos, ref, ref[:1] + rand_seq(rng, rng.randrange(1, 6)), "Insertion", rng.choice(ZY))
            if kind == 2:
                return V(pos, ref, ref[:1], "Deletion", rng.choice(ZY))
            return V(pos, ref, rand_seq(rng, rng.randrange(2, 6)), "Indel", rng.choice(ZY))
        truth = sorted([rv() for _ in range(rng.randrange(0, 4))], key=lambda v: v["pos"])
        if truth and rng.random() < 0.6:
            query = [dict(v, zyg=(v["zyg"] if rng.random() < 0.7 else rng.choice(ZY))) for v in truth if rng.random() > 0.2]
            query += [rv() for _ in range(rng.randrange(0, 2))]
        else:
            query = [rv() for _ in range(rng.randrange(0, 4))]
        query.sort(key=lambda v: v["pos"])
        if not truth and not query:
            continue
        regions.append(dict(start=start, end=end, truth=truth, query=query))
    out.append(dict(name="random_small", contig=contig4, regions=regions, max_branch_factor=50))

    # 5. (round 4) clusters of unphased heterozygous call pairs on short windows: symmetric phasing searches with up to dozens of tied optima, whose ORDER decides the
    # winner (waffle_solver.rs:264-265) — what the wave-cooperative kernel expands sixteen queue entries at a time and commits in the reference's order
    r5 = random.Random(20251003)
    contig5 = rand_seq(r5, 2400)
    regions5 = []
    for k in range(14):
        L = r5.randrange(90, 200)
        start = 20 + 165 * k
        end = start + L
        n_sites = 3 + k % 4
        pos = sorted(r5.sample(range(start + 3, end - 8), n_sites))
        truth, query = [], []
        for p in pos:
            kind = r5.random()
            if kind < 0.75:
                a0, a1, vt = contig5[p:p + 1], bytes([other_base(contig5[p], r5)]), "Snv"
            elif kind < 0.88:
                a0, a1, vt = contig5[p:p + 1], contig5[p:p + 1] + rand_seq(r5, r5.randrange(1, 5)), "Insertion"
            else:
                a0, a1, vt = contig5[p:p + r5.randrange(2, 6)], contig5[p:p + 1], "Deletion"
            zt = "UnphasedHeterozygous" if r5.random() < 0.8 else r5.choice(ZY)
            zq = "UnphasedHeterozygous" if r5.random() < 0.85 else r5.choice(ZY)
            if r5.random() > 0.08:
                truth.append(V(p, a0, a1, vt, zt))
            if r5.random() > 0.08:
                qa1 = bytes([other_base(contig5[p], r5)]) if vt == "Snv" and r5.random() < 0.12 else a1  # sometimes another ALT base on the query side
                query.append(V(p, a0, qa1, vt, zq))
        if k % 5 == 4 and truth:  # a multi-allelic site: two records at one position on each side (the second is skipped on the haplotype that took the first)
            v = truth[0]
            if v["type"] == "Snv":
                third = bytes([c for c in b"ACGT" if c != v["a0"][0] and c != v["a1"][0]][:1])
                truth.insert(1, V(v["pos"], v["a0"], third, "Snv", "UnphasedHeterozygous"))
                query.insert(0, V(v["pos"], v["a0"], third, "Snv", "UnphasedHeterozygous"))
                query.sort(key=lambda x: x["pos"])
        regions5.append(dict(start=start, end=end, truth=truth, query=query))
    for mb in (50, 3):
        out.append(dict(name="het_clusters_mbf%d" % mb, contig=contig5, regions=regions5, max_branch_factor=mb))

    # 6. (round 4) one SNV per side, the same one — the class that is looked up, not searched (avk_pairs.inl): every pair of zygosities, with the window cut by the
    # contig's start and end and the call on the window's first and last base; branch factors 1, 2, 3 (the quota reaches a two-call search at 1 and 2)
    r6 = random.Random(616)
    contig6 = rand_seq(r6, 260)
    regions6 = []
    spots = [(0, 40, 0), (0, 40, 39), (0, 51, 25), (len(contig6) - 45, len(contig6), len(contig6) - 1), (len(contig6) - 45, len(contig6), len(contig6) - 45), (100, 201, 150)]
    for i, (start, end, p) in enumerate(spots):
        for j, (zt, zq) in enumerate([(a, b) for a in ZY for b in ZY]):
            if (i + j) % 3 and i not in (0, 3):
                continue  # all sixteen pairs at the two clipped spots, a third of them elsewhere
            alt = bytes([other_base(contig6[p], r6)])
            regions6.append(dict(start=start, end=end, truth=[V(p, contig6[p:p + 1], alt, "Snv", zt)], query=[V(p, contig6[p:p + 1], alt, "Snv", zq)]))
    for mb in (1, 2, 3):
        out.append(dict(name="same_snv_edges_mbf%d" % mb, contig=contig6, regions=regions6, max_branch_factor=mb))

    # 7. (round 4) records whose REF allele is not what the genome has at their position (the solver splices the ALT into the WINDOW, waffle_solver.rs:726-778, while
    # Variant::alt_ed and the skip penalty come from the record's own alleles): an SNV whose ALT is the genome's base, an SNV with a foreign REF, indels written with
    # another anchor base, against plain calls at the same place
    r7 = random.Random(717)
    contig7 = rand_seq(r7, 1600)
    regions7 = []
    for k in range(16):
        start = 30 + 95 * k
        end = start + r7.randrange(60, 90)
        p = r7.randrange(start + 5, end - 12)
        g = contig7[p:p + 1]
        wrong = bytes([other_base(contig7[p], r7)])
        wrong2 = bytes([c for c in b"ACGT" if c != contig7[p] and c != wrong[0]][:1])
        ins = rand_seq(r7, r7.randrange(1, 4))
        dl = r7.randrange(2, 5)
        shapes = [
            V(p, wrong, g, "Snv", r7.choice(ZY)),                                  # "ALT" is the genome's base: the haplotype IS the window
            V(p, wrong, wrong2, "Snv", r7.choice(ZY)),                             # foreign REF, foreign ALT
            V(p, wrong, wrong + ins, "Insertion", r7.choice(ZY)),                  # insertion behind an anchor the genome does not have
            V(p, wrong + contig7[p + 1:p + dl], wrong, "Deletion", r7.choice(ZY)),  # deletion with a foreign anchor
            V(p, g + contig7[p + 1:p + dl], wrong, "Deletion", r7.choice(ZY)),      # deletion that also changes the anchor
        ]
        plain = [V(p, g, wrong, "Snv", r7.choice(ZY)), V(p, g, g + ins, "Insertion", r7.choice(ZY)), V(p, contig7[p:p + dl], g, "Deletion", r7.choice(ZY))]
        t = [shapes[k % 5]]
        q = [r7.choice(plain)] if k % 3 else [dict(shapes[k % 5], zyg=r7.choice(ZY))]
        if k % 4 == 0:  # a second, ordinary call further on
            p2 = p + dl + r7.randrange(3, 8)
            if p2 < end - 2:
                extra = V(p2, contig7[p2:p2 + 1], bytes([other_base(contig7[p2], r7)]), "Snv", r7.choice(ZY))
                t.append(extra)
                q.append(dict(extra, zyg=r7.choice(ZY)))
        regions7.append(dict(start=start, end=end, truth=t, query=q))
    out.append(dict(name="ref_allele_disagrees", contig=contig7, regions=regions7, max_branch_factor=50))
    return out


# ---------------------------------------------------------------- merge_solver.rs:110-223 (round 4: four and five inputs)
def variant_delta_length(variants):  # :211-223
    total = 0
    for v in variants:
        if v["zyg"] == "Unknown":
            raise SolverError("BAD_ZYGOSITY")
        total += (len(v["a1"]) - len(v["a0"])) * zyg_count(v["zyg"])
    return total


def solve_merge_region(reference, region, no_conflict_enabled, majority_voting_enabled, conflict_selection, max_branch=50):
    inputs = region["inputs"]
    k = len(inputs)
    delta = [variant_delta_length(v) for v in inputs]
    all_identical, no_conflict = True, True
    match_sets = [set([i]) for i in range(k)]
    for i in range(k):
        for j in range(i + 1, k):
            if delta[i] == delta[j]:
                best = optimize_sequences(reference, region["start"], region["end"], inputs[i], [v["zyg"] for v in inputs[i]], inputs[j], [v["zyg"] for v in inputs[j]], max_branch)[0]
                exact = best["ed1"] == 0 and best["ed2"] == 0 and best["tvs1"] == 0 and best["tvs2"] == 0 and best["qvs1"] == 0 and best["qvs2"] == 0  # is_exact_match, query_optimizer.rs:96-98
            else:
                exact = False
            all_identical = all_identical and exact
            no_conflict = no_conflict and (not inputs[i] or not inputs[j] or exact)
            if exact:
                match_sets[i].add(j)
                match_sets[j].add(i)
    maj_count = k // 2 + 1
    first_maj = next((sorted(m) for m in match_sets if len(m) >= maj_count), [])
    if all_identical:
        return ["BasepairIdentical"]
    if no_conflict_enabled and no_conflict:
        return ["NoConflict", [i for i, v in enumerate(inputs) if v]]
    if majority_voting_enabled and first_maj:
        return ["MajorityAgree", first_maj]
    if conflict_selection is not None:
        return ["ConflictSelection", conflict_selection]
    return ["Different"]


def merge_scenarios():
    r = random.Random(20251004)
    contig = rand_seq(r, 3000)
    ZY = ["UnphasedHeterozygous", "PhasedHet01", "PhasedHet10", "HomozygousAlternate"]
    regions = []
    for n in range(36):
        k = 4 + n % 2
        L = r.randrange(50, 140)
        start = 10 + 80 * n
        end = start + L

        def rv():
            kind = r.randrange(3)
            p = r.randrange(start + 2, end - 8)
            if kind == 0:
                return V(p, contig[p:p + 1], bytes([other_base(contig[p], r)]), "Snv", r.choice(ZY))
            if kind == 1:
                return V(p, contig[p:p + 1], contig[p:p + 1] + rand_seq(r, r.randrange(1, 4)), "Insertion", r.choice(ZY))
            return V(p, contig[p:p + r.randrange(2, 5)], contig[p:p + 1], "Deletion", r.choice(ZY))
        base = sorted([rv() for _ in range(r.randrange(1, 4))], key=lambda v: v["pos"])

        def unphase(vs):  # the same calls written by another caller: phase dropped (still an exact match: the search finds the orientation)
            return [dict(v, zyg="UnphasedHeterozygous" if v["zyg"].startswith("PhasedHet") and r.random() < 0.7 else v["zyg"]) for v in vs]
        inputs = []
        for i in range(k):
            u = r.random() if n % 6 else 0.0  # (every sixth region: all callers agree)
            if u < 0.45:
                inputs.append(unphase(base))
            elif u < 0.6:
                inputs.append([])                                   # a caller without calls here
            elif u < 0.8:
                other = [dict(v) for v in base]
                w = r.randrange(len(other))
                other[w] = dict(other[w], zyg=r.choice([z for z in ZY if zyg_count(z) != zyg_count(other[w]["zyg"])] or ZY))  # another genotype
                inputs.append(other)
            else:
                inputs.append(sorted([rv() for _ in range(r.randrange(1, 3))], key=lambda v: v["pos"]))  # unrelated calls
        regions.append(dict(start=start, end=end, inputs=inputs))
    configs = [dict(no_conflict_enabled=False, majority_voting_enabled=False, conflict_selection=None), dict(no_conflict_enabled=False, majority_voting_enabled=True, conflict_selection=None),
               dict(no_conflict_enabled=True, majority_voting_enabled=True, conflict_selection=None), dict(no_conflict_enabled=True, majority_voting_enabled=False, conflict_selection=2)]
    return contig, regions, configs


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    fixtures = []
    for sc in scenarios():
        stats = {}
        expect = []
        for r in sc["regions"]:
            try:
                expect.append(solve_compare_region(sc["contig"], r, sc["max_branch_factor"], stats))
            except SolverError as e:
                expect.append(dict(status=str(e)))
        print("%-22s %d regions, max_branch_factor %d: statuses %s; phasing search: nodes dropped by the quota %d; gt search: most expansions %d, auto-fail prunings %d" %
              (sc["name"], len(sc["regions"]), sc["max_branch_factor"], sorted(set(x["status"] for x in expect)), stats.get("quota_drops", 0), stats.get("max_expansions", 0),
               stats.get("autofail", 0)))
        fixtures.append(dict(name=sc["name"], contig=sc["contig"].decode("latin1"), max_branch_factor=sc["max_branch_factor"],
                             generator_stats=dict(quota_drops=stats.get("quota_drops", 0), gt_max_expansions=stats.get("max_expansions", 0), autofail_prunings=stats.get("autofail", 0)),
                             regions=[dict(start=r["start"], end=r["end"],
                                           truth=[[v["pos"], v["a0"].decode("latin1"), v["a1"].decode("latin1"), v["type"], v["zyg"], v["raw"]] for v in r["truth"]],
                                           query=[[v["pos"], v["a0"].decode("latin1"), v["a1"].decode("latin1"), v["type"], v["zyg"], v["raw"]] for v in r["query"]]) for r in sc["regions"]],
                             expect=expect))
    json.dump(dict(source="tests/golden/make_crosscheck.py: an independent Python restatement of the reference solver (see its header for the lines it follows)", scenarios=fixtures),
              open(os.path.join(here, "crosscheck.json"), "w"), indent=None, separators=(",", ":"))
    print("wrote crosscheck.json")
    contig, regions, configs = merge_scenarios()
    out = []
    for cfg in configs:
        out.append(dict(config=cfg, expect=[solve_merge_region(contig, reg, **cfg) for reg in regions]))
    kinds = sorted(set(e[0] for c in out for e in c["expect"]))
    print("merge: %d regions of 4 / 5 inputs x %d configurations: classifications %s" % (len(regions), len(configs), kinds))
    json.dump(dict(source="tests/golden/make_crosscheck.py: solve_merge_region (merge_solver.rs:110-223) restated on top of the same script's optimize_sequences", contig=contig.decode("latin1"),
                   regions=[dict(start=reg["start"], end=reg["end"], inputs=[[[v["pos"], v["a0"].decode("latin1"), v["a1"].decode("latin1"), v["type"], v["zyg"], v["raw"]] for v in inp]
                                                                             for inp in reg["inputs"]]) for reg in regions], cases=out),
              open(os.path.join(here, "merge_crosscheck.json"), "w"), indent=None, separators=(",", ":"))
    print("wrote merge_crosscheck.json")


if __name__ == "__main__":
    main()
