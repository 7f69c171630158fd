/*
 * avk_quad.inl — the compare solver for the EXPENSIVE small regions: one region per QUAD (four lanes), sixteen regions per wavefront.
 *
 * The heads of the lane classes (regions with estimated edits) and the three-call class are where a step's vector instructions go
 * (52 % of them on 2 % of a genome's regions), and avk_lane.inl runs them sixteen records per wave with the other 48 lanes switched off:
 * lanes that diverge take turns, so narrow tiles on many waves beat wide ones.  A region's work is not one chain, though — at every level
 * of solve_compare_region (src/waffle_solver.rs:122-284) there are four independent pieces:
 *
 *   - optimize_sequences (src/query_optimizer.rs:203-328): a popped node is extended on haplotype 1 and haplotype 2, and for an unphased
 *     heterozygous call into two children — lane (child, haplotype) runs ONE HaplotypeDWFA::extend_variant (src/dwfa/haplotype_dwfa.rs:46-67);
 *   - optimize_gt_alleles (src/exact_gt_optimizer.rs:108-357) is one search per haplotype of an optimum;
 *   - add_basepair_stats (src/waffle_solver.rs:335-449): ed(ref, truth), ed(ref, query) per haplotype — lane (side, haplotype) — and the
 *     filtered alignments per call type, again one per (side, haplotype); the metric groups, one per lane.
 *
 * So the four lanes of a quad share one region's tables (the rows of avk_lane.inl, [word][region] with a stride of sixteen), each lane owns
 * one alignment front, and what the lanes of a quad need from each other travels over DPP quad permutes (avk_wave.h: qd_xor, qd_bcast).
 * Everything that decides a result — queue words, pop order, node ids, the per-depth quota, lazily evaluated costs, the order of tied
 * optima — is avk_lane.inl's, i.e. the reference's: the search is the same search, its independent pieces side by side.  The primitives
 * (2-bit sequences, the row-wise wavefront aligner, haplotype steps, the metrics phase's forced distances) are avk_lane.inl's templates on
 * the same LCtx.
 *
 * Control flow is uniform over a quad and divergent over the wave; a lane that must give up (LS_DEFER) carries the flag to the quad's next
 * exchange, where all four leave together.  Shared rows are written by ONE lane of the quad (the queue and the optima list by lane 0, a kept
 * node state by the lanes that hold it) with a quad primitive between any write and another lane's read.
 */
#ifndef AVK_QUAD_INL
#define AVK_QUAD_INL

#include "avk_lane.inl"

namespace avk {
namespace quad {

using namespace avk::lane;

/* profiling builds (-DAVK_LANE_PHASE_TIMING -DAVK_QUAD_FINE=1 or 2): the clock ticks of the staging (1) or of the phasing search (2) split further; the marks
 * take the place of the phases' own, everything else lands in slot 0 (tools/gpu_lane_phases.py prints the slots under their fine names) */
#if defined(AVK_LANE_PHASE_TIMING) && defined(AVK_QUAD_FINE)
#define AVK_QF(which, c, k)                   \
    if (AVK_QUAD_FINE == which) {             \
        const u64 n_ = avk_clock();           \
        (c).tph[k] += n_ - (c).tlast;         \
        (c).tlast = n_;                       \
    }
#undef AVK_LT_MARK
#define AVK_LT_MARK(c, k) AVK_QF(AVK_QUAD_FINE, c, 0)
#else
#define AVK_QF(which, c, k)
#endif

/* rows of a region (stride = regions per wave):
 *   sequence table | optima list | DYN = queue, kept node states (pool x 2 haplotypes x wfr), the four lanes' search fronts (4 x wfr)
 * after the search DYN is four private arrays of DYN / 4 rows: a lane's front while an optimum is replayed, its queue in the genotype search,
 * its one long array in the alignments of the metrics phase */
AVK_DEV u32 quad_rows(u32 W, u32 nm, u32 ed_max, u32 qcap, u32 pool) {
    const u32 ns = 1 + 2 * (nm - 1);
    const u32 wfr = (2 * ed_max + 2 + 3) / 4;
    return ns * (W + 1) + lane_optcap(nm) / 2 + qcap + (2 * pool + 4) * wfr;
}

/* summary of a node's two haplotypes after an extension step, the same word on the node's two lanes */
enum { QS_PARTIAL = 1u << 24, QS_DEFER = 1u << 25 };
AVK_DEV u32 qs_cost(u32 s) { return s & 0xFFFFu; }
AVK_DEV u32 qs_eds(u32 s) { return (s >> 16) & 0xFFu; } /* ed of haplotype 0 | ed of haplotype 1 << 4 */

/* ---- one haplotype of a node, from the node's path ------------------------------------------------------------------- */
AVK_DEV void hapq_replay_steps(const LCtx &c, Hap &h, u32 k, u32 code, u32 depth) { /* lengths, positions, skipped calls */
    hap_init(h);
    for (u32 d = 0; d < depth; ++d) {
        const u32 slot = ord_slot(c, d), choice = (code >> (2 * d)) & 3u;
        hap_step(c, h, slot < MV, true, slot, ((choice >> k) & 1u) ? L_ALT : L_REF, sync_after(c, d));
    }
}
/* nodeA_restore_zero for haplotype k: a node of cost 0 in a region without zero-distance calls, in closed form */
AVK_DEV void hapq_restore_zero(const LCtx &c, Hap &h, u32 k, u32 code, u32 depth) {
    u32 alt[2] = {0, 0}, cnt[2] = {0, 0};
    for (u32 d = 0; d < depth; ++d) {
        const u32 b = (code >> (2 * d + k)) & 1u, q = (c.takeq >> d) & 1u;
        const u32 at = q ? cnt[1] : cnt[0];
        alt[0] |= q ? 0u : b << at;
        alt[1] |= q ? b << at : 0u;
        cnt[0] += 1u - q, cnt[1] += q;
    }
    const u32 sync = depth ? sync_after(c, depth - 1) : 0u;
    u32 pos[2], len[2];
#pragma unroll
    for (u32 side = 0; side < 2; ++side) {
        const u32 m = alt[side];
        const u32 last = m ? (c.ends[side] >> (10u * (31u - (u32)__builtin_clz(m)))) & 0x3FFu : 0u;
        pos[side] = last > sync ? last : sync;
        len[side] = pos[side] + c.seq_len(c.seq_id(side, m)) - c.L;
    }
    h.t_refpos = pos[0], h.q_refpos = pos[1], h.t_len = len[0], h.q_len = len[1];
    h.t_skip = h.q_skip = h.nskip = 0;
    h.t_alt = alt[0], h.q_alt = alt[1], h.t_nal = cnt[0], h.q_nal = cnt[1];
    h.ed = 0;
    h.d0 = len[0] < len[1] ? len[0] : len[1];
}
AVK_DEV void hapq_replay_zero(const LCtx &c, Hap &h, u32 k, u32 code, u32 depth) {
    if (c.plain) {
        hapq_restore_zero(c, h, k, code, depth);
        return;
    }
    hapq_replay_steps(c, h, k, code, depth);
    h.d0 = h.t_len < h.q_len ? h.t_len : h.q_len;
}
/* the whole path again, every step aligned exactly (nodeA_replay for one haplotype); true: give up */
AVK_DEV bool hapq_replay_exact(const LCtx &c, Hap &h, u32 k, u32 code, u32 depth) {
    hap_init(h);
    for (u32 d = 0; d < depth; ++d) {
        const u32 slot = ord_slot(c, d), choice = (code >> (2 * d)) & 3u;
        hap_step(c, h, slot < MV, true, slot, ((choice >> k) & 1u) ? L_ALT : L_REF, sync_after(c, d));
        if (hap_update(c, h, 0)) return true;
    }
    return false;
}

/* ---- kept node states (NodePool of avk_lane.inl): the bookkeeping is the same on the four lanes, a haplotype's front is written by the lane that holds it */
AVK_DEV u32 *qpool_rows(const LCtx &c, u32 s, u32 k) { return c.p + ((c.off_pool + (2 * s + k) * c.wfr) << c.ls); }
AVK_DEV void qpool_load(const LCtx &c, NodePool &pl, u32 s, u32 k, Hap &h) { /* the haplotype steps of h are made; frees the slot */
    const u32 e = (u32)(pl.eds >> (8 * s)) & 0xFFu;
    h.ed = k ? e >> 4 : e & 15u;
    const u32 *src = qpool_rows(c, s, k);
    if (h.ed == 0) h.d0 = src[0] & 0xFFu;
    else {
        const u32 rows = (2 * h.ed + 1 + 3) >> 2;
        for (u32 r = 0; r < rows; ++r) *c.wf_row(0, r) = src[r << c.ls];
    }
    pool_free(pl, s);
}
/* a queued node's state goes into a free slot, if there is one and the path alone does not say it all; `writer`: this lane holds haplotype k of that node */
AVK_DEV void qpool_keep(const LCtx &c, NodePool &pl, u32 id, const Hap &h, u32 k, u32 eds, bool partial, bool writer) {
    if (!c.pool || !(partial || eds)) return;
    u32 s = c.pool;
    for (u32 j = 0; j < c.pool; ++j)
        if (((u32)(pl.ids >> (8 * j)) & 0xFFu) == 0xFFu) s = j;
    if (s >= c.pool) return;
    pl.ids = (pl.ids & ~(0xFFull << (8 * s))) | ((u64)id << (8 * s));
    pl.eds = (pl.eds & ~(0xFFull << (8 * s))) | ((u64)eds << (8 * s));
    if (!writer) return;
    u32 *dst = qpool_rows(c, s, k);
    if (h.ed == 0) dst[0] = h.d0;
    else {
        const u32 rows = (2 * h.ed + 1 + 3) >> 2;
        for (u32 r = 0; r < rows; ++r) dst[r << c.ls] = *c.wf_row(0, r);
    }
}

/* ---- the queue: lane 0's rows, its pops broadcast ------------------------------------------------------------------- */
AVK_DEV int qq_push(const LCtx &c, u32 q, u32 &qn, u32 e) {
    if (qn >= c.qcap) return AVK_LDEFER(1);
    if (q == 0) c.p[(c.off_q + qn) << c.ls] = e;
    qn += 1;
    return 0;
}
AVK_DEV int qq_pushA(const LCtx &c, u32 q, u32 &qn, u32 cost, u32 id, u32 code, u32 partial, u32 depth) {
    if (cost > 255u) return AVK_LDEFER(5);
    return qq_push(c, q, qn, keyA(cost, id, code, partial, depth));
}
/* the smallest queue word, removed; `second` = the smallest of what is left (only made for partial entries).  The four lanes scan a quarter of the queue each. */
AVK_DEV u32 qd_min(u32 v) {
    const u32 a = qd_xor(1, v);
    v = a < v ? a : v;
    const u32 b = qd_xor(2, v);
    return b < v ? b : v;
}
AVK_DEV u32 qq_pop_min(const LCtx &c, u32 q, u32 &qn, u32 &second) {
    qd_sync(); /* lane 0's pushes */
    u32 best = 0xFFFFFFFFu, sec = 0xFFFFFFFFu, bi = 0;
    for (u32 i = q; i < qn; i += 4) {
        const u32 e = c.p[(c.off_q + i) << c.ls];
        if (e < best) {
            sec = best;
            best = e;
            bi = i;
        } else if (e < sec)
            sec = e;
    }
    const u32 gbest = qd_min(best);
    const u32 gi = qd_min(best == gbest ? bi : 0xFFFFFFFFu); /* (queue words are unique: one lane holds it) */
    const u32 last = c.p[(c.off_q + qn - 1) << c.ls];
    qd_sync(); /* every lane has read the queue */
    if (q == 0) c.p[(c.off_q + gi) << c.ls] = last;
    qn -= 1;
    second = 0xFFFFFFFFu;
    if (gbest & 8u) second = qd_min(best == gbest ? sec : best);
    return gbest;
}

/* The alignments of a node's last step (nodeA_settle / finalize_dwfas), one haplotype per lane, in two halves around the ONE call of the aligner the search
 * has (phaseA_quad).  `cap` = the largest total node cost the caller cares about.  The budget of a haplotype is what the cap leaves beside the skipped calls
 * and the OTHER haplotype's distance before this step — avk_lane.inl aligns the second haplotype with what the first one left, which only stops it earlier:
 * either way a node comes back partial with a lower bound that is not above its cost, or exact, and the sequence of real pops is the reference's. */
struct Settle {
    u32 base, oed, lb0;
    bool defer0, over;
};
AVK_DEV void settle_before(const Hap &h, u32 cap, bool defer_in, Settle &t) {
    const u32 mine0 = (h.t_skip + h.q_skip) | (h.ed << 16) | (defer_in ? 1u << 24 : 0u);
    const u32 oth0 = qd_xor(1, mine0);
    t.base = (mine0 & 0xFFFFu) + (oth0 & 0xFFFFu), t.oed = (oth0 >> 16) & 0xFFu;
    t.defer0 = (((mine0 | oth0) >> 24) & 1u) != 0;
    t.lb0 = t.base + h.ed + t.oed; /* distances never decrease */
    t.over = t.lb0 > cap;
}
/* the node's summary, the same on its two lanes: cost or lower bound | eds << 16 | QS_PARTIAL | QS_DEFER */
AVK_DEV u32 settle_after(const Hap &h, u32 k, u32 cap, int r, const Settle &t) {
    const u32 mine1 = h.ed | (r == LS_PARTIAL ? 1u << 8 : 0u) | (r == LS_DEFER ? 1u << 9 : 0u);
    const u32 oth1 = qd_xor(1, mine1);
    const bool defer = t.defer0 || (((mine1 | oth1) >> 9) & 1u);
    const bool partial = t.over || (((mine1 | oth1) >> 8) & 1u);
    const u32 ed_me = h.ed, ed_ot = oth1 & 0xFFu;
    const u32 eds = k ? (ed_ot | (ed_me << 4)) : (ed_me | (ed_ot << 4));
    u32 cst = partial ? (t.over ? t.lb0 : cap + 1u) : t.base + ed_me + ed_ot;
    cst = cst < 0xFFFFu ? cst : 0xFFFFu;
    return cst | (eds << 16) | (partial ? (u32)QS_PARTIAL : 0u) | (defer ? (u32)QS_DEFER : 0u);
}

/* ---- phase A: optimize_sequences (avk_lane.inl phaseA, one haplotype per lane, two children side by side) ------------
 * q = lane of the quad: haplotype k = q & 1, child ch = q >> 1.  Returns the number of tied optima, LS_DEFER, or -100 - status.
 *
 * BRANCH-FLATTENED: the sixteen quads of a wave go round this loop together, whatever each of them popped — a partial entry to be taken further, a finished
 * node, a node that splits in two or moves on — so the loop body has ONE haplotype step, ONE aligner and ONE push / keep sequence, and what differs between
 * the cases is data: from which depth the node's path is walked again (`d`), from where its steps are aligned (`d_align`), which allele the last step takes,
 * what the cap is and whether the last alignment is a finalisation.  (Sixteen quads in five copies of the aligner take five turns.)
 * A partial entry that turns out to cost exactly what it was popped for goes back into the queue as an exact entry — the smallest word there, so it is the
 * next pop, the real one, with its state kept — instead of being expanded in the same round. */
AVK_DEV int phaseA_quad(const LCtx &c, u32 q, u32 &best_out) {
    const u32 k = q & 1u, ch = q >> 1;
    u32 qn = 0;
    qq_push(c, q, qn, 0);
    u32 next_id = 1, best = 0xFFFFu, nbest = 0;
    u64 bucket = 0;
    NodePool pl;
    pl.ids = ~0ull, pl.eds = 0;
    while (qn > 0) {
        u32 second;
        const u32 e = qq_pop_min(c, q, qn, second);
        AVK_QF(2, c, 1)
        const u32 cost = e >> 24;
        if (cost > best) break; /* :204 */
        const u32 depth = e & 7u, code = (e >> 4) & 0xFFFu, id = (e >> 16) & 0xFFu;
        const bool part = (e & 8u) != 0;
        /* ---- the plan of this round */
        u32 cap, d_to, choice = 0;
        bool two = false;
        if (part) { /* the last step is taken further: as far as it can matter for the order (the next entry's cost), at least doubling, never beyond the best finished cost */
            cap = qn ? second >> 24 : 0xFFFFu;
            cap = cap > 2 * cost + 2 ? cap : 2 * cost + 2;
            cap = cap < best ? cap : best;
            cap = cap > cost ? cap : cost;
            d_to = depth - 1;
        } else {
            const u32 cnt = (u32)(bucket >> (8 * depth)) & 0xFFu;
            if (cnt >= c.max_branch) { /* :222 */
                if (cost) {
                    const u32 slot = pool_find(c, pl, id);
                    if (slot < c.pool) pool_free(pl, slot);
                }
                continue;
            }
            bucket += 1ull << (8 * depth);
            d_to = depth;
            cap = cost;
            if (depth == c.N) cap = best; /* :227-247; finalize_dwfas (:457-462) */
            else {
                const u32 slot = ord_slot(c, depth);
                const u32 zyg = (sel4(c.vw0, slot) >> 28) & 7u;
                const bool het = zyg == AVK_ZYG_UNPHASED_HET || zyg == AVK_ZYG_PHASED_HET01 || zyg == AVK_ZYG_PHASED_HET10;
                two = het && (slot >= MV || zyg == AVK_ZYG_UNPHASED_HET); /* :269-293: two clones, (REF|ALT) then (ALT|REF): child 0 = choice 2, child 1 = choice 1 */
                choice = two ? (ch ? 1u : 2u) : (het ? (zyg == AVK_ZYG_PHASED_HET01 ? 2u : 1u) : 3u); /* :294-327: the node is moved (the second pair of lanes does what the first does) */
            }
        }
        const bool final = !part && depth == c.N;
        /* ---- the node's state: in closed form, or its path walked again (lengths only up to the last step; with every step aligned when nothing else is left) */
        Hap h;
        bool dfr = false, need_front = false;
        const u32 kept = (part || cost) ? pool_find(c, pl, id) : c.pool;
        u32 d = 0, d_align = d_to;
        if (!part && cost == 0 && c.plain) {
            hapq_restore_zero(c, h, k, code, depth);
            d = depth;
        } else {
            hap_init(h);
            need_front = true;
            if (part && kept >= c.pool) d_align = 0, need_front = false; /* where the step stopped was not kept */
        }
        Settle t;
        t.base = t.oed = t.lb0 = 0, t.defer0 = t.over = false;
        int r_last = 0;
        for (;;) {
            const bool last = d == d_to;
            bool load_kept = false;
            if (last && need_front) { /* the state in front of the last step is complete but for where its alignments stand */
                need_front = false;
                load_kept = part; /* (a partial entry's kept fronts are those of its last step) */
                if (!part) {
                    const u32 sk = h.t_skip + h.q_skip;
                    if (sk + qd_xor(1, sk) == cost) h.d0 = h.t_len < h.q_len ? h.t_len : h.q_len; /* all skipped calls: the fronts are the ends of the shorter sequences */
                    else if (kept < c.pool) load_kept = true;
                    else { /* no kept state: the path again, every step aligned */
                        hap_init(h);
                        d = 0, d_align = 0;
                        continue;
                    }
                }
            }
            /* the step of depth d */
            bool is_truth = true, has_var = false;
            u32 slot = 0, allele = L_REF, sync = c.L;
            if (d < c.N) {
                slot = ord_slot(c, d);
                is_truth = slot < MV;
                has_var = true;
                sync = sync_after(c, d);
                const u32 chc = d == depth ? choice : (code >> (2 * d)) & 3u;
                allele = ((chc >> k) & 1u) ? L_ALT : L_REF;
            }
            hap_step(c, h, is_truth, has_var, slot, allele, sync);
            if (load_kept) qpool_load(c, pl, kept, k, h); /* (a step changes lengths and positions, not the fronts) */
            if (d >= d_align) {
                u32 budget = 0xFFFFu;
                bool run = !dfr;
                if (last) {
                    settle_before(h, cap, dfr, t);
                    run = !t.defer0 && !t.over;
                    budget = cap - t.base - t.oed;
                }
                int r = 0;
                if (run) r = hap_align(c, h, 0, budget, last && final);
                if (last) r_last = r;
                else dfr = dfr || r != 0;
            }
            if (last) break;
            d += 1;
        }
        const u32 s = settle_after(h, k, cap, r_last, t);
        AVK_QF(2, c, 3)
        if (final) {
            if (s & QS_DEFER) return LS_DEFER;
            if (s & QS_PARTIAL) continue; /* costs more than the best: neither kept nor tied */
            const u32 fc = qs_cost(s);
            if (fc < best) {
                best = fc;
                nbest = 0;
            }
            if (fc == best) {
                if (nbest >= c.optcap) return AVK_LDEFER(2);
                if (q == 0) opt_set(c, nbest, code);
                nbest += 1;
            }
            continue;
        }
        /* ---- what goes (back) into the queue: a partial entry as it is now, a moved node, or two children */
        const u32 o = two ? qd_xor(2, s) : s;
        const u32 s0 = (two && ch) ? o : s, s1 = ch ? s : o;
        if ((s0 | (two ? s1 : 0u)) & QS_DEFER) return LS_DEFER;
        const u32 n_push = two ? 2u : 1u;
        for (u32 j = 0; j < n_push; ++j) {
            const u32 sj = j ? s1 : s0;
            const u32 pid = part ? id : (two ? next_id + j : id);
            const u32 pdepth = part ? depth : depth + 1;
            const u32 pcode = part ? code : code | ((two ? (j ? 1u : 2u) : choice) << (2 * depth));
            if (qq_pushA(c, q, qn, qs_cost(sj), pid, pcode, (sj & QS_PARTIAL) ? 1u : 0u, pdepth)) return LS_DEFER;
            qpool_keep(c, pl, pid, h, k, qs_eds(sj), (sj & QS_PARTIAL) != 0, ch == j);
        }
        if (two) next_id += 2;
        AVK_QF(2, c, 4)
        if (next_id > c.max_nodes) return AVK_LDEFER(3);
    }
    if (nbest == 0) return -100 - AVK_ST_NO_RESULTS; /* :331 */
    best_out = best;
    return (int)nbest;
}

/* OR over the quad of a 64-bit value */
AVK_DEV u64 qd_or64(u64 v) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo |= qd_xor(1, lo);
    hi |= qd_xor(1, hi);
    lo |= qd_xor(2, lo);
    hi |= qd_xor(2, hi);
    return ((u64)hi << 32) | lo;
}
AVK_DEV bool qd_any(bool p) {
    u32 v = p ? 1u : 0u;
    v |= qd_xor(1, v);
    v |= qd_xor(2, v);
    return v != 0;
}

/* what the metrics phase reads of the winner's two haplotypes, on every lane */
struct WinHap {
    u32 t_alt, q_alt, o_t, o_q, ed, nskip, t_skip, q_skip, t_len, q_len;
};
AVK_DEV void win_unpack(WinHap &w, u32 a, u32 b) {
    w.t_alt = a & 7u, w.q_alt = (a >> 3) & 7u, w.o_t = (a >> 6) & 7u, w.o_q = (a >> 9) & 7u, w.ed = (a >> 12) & 0xFFu, w.nskip = (a >> 20) & 0xFu;
    w.t_skip = b & 0xFFu, w.q_skip = (b >> 8) & 0xFFu, w.t_len = (b >> 16) & 0xFFu, w.q_len = b >> 24;
}

/* returns AVK_ST_* (>= 0) or LS_DEFER, the same value on the four lanes of the quad.  `dyn_rows` = rows of the region's DYN area (quad_rows). */
AVK_DEV int solve_quad(const AvkKernelArgs &a, LCtx &c, u32 q, const u32 *rec, u32 lane_stride, u32 dyn_rows, u32 max_ed_c, LaneOut &out, u32 *tally) {
    const u32 h0 = rec[0], h1 = rec[1 * lane_stride], v_off = rec[2 * lane_stride], orig = rec[3 * lane_stride];
    const u32 shift = h1 & 15u;
    c.L = (h1 >> 4) & 0xFFu;
    c.T = (h1 >> 12) & 3u;
    c.Q = (h1 >> 14) & 3u;
    c.N = c.T + c.Q;
    {
        const u32 takeq = h1 >> 16;
        u32 ord = 0, it = 0, iq = 0;
        for (u32 d = 0; d < c.N; ++d) {
            const u32 qq = (takeq >> d) & 1u;
            ord |= (qq ? MV + iq : it) << (3 * d);
            iq += qq;
            it += 1u - qq;
        }
        c.ord = ord;
        c.takeq = takeq & ((1u << c.N) - 1u);
    }
    c.max_branch = a.max_branch_factor;
    c.seq_len_lo = c.L;
    c.seq_len_hi = c.seq_fail_lo = c.seq_fail_hi = 0;
    u32 a1lo[NS], a1hi[NS];
    u32 types = 0;
    const u32 maxv = c.nm1 == 1 ? 1u : (c.nm1 == 3 ? 2u : 3u);
#pragma unroll
    for (u32 s = 0; s < NS; ++s) { /* every slot the class's records have is loaded, whatever the header says: one round trip for the whole record */
        const u32 side = s / MV, j = s % MV;
        const bool on = j < (side ? c.Q : c.T);
        c.vw0[s] = c.vw1[s] = a1lo[s] = a1hi[s] = 0;
        if (j < maxv) {
            const u32 *v = rec + (AVK_FAST_HDR + 4 * (side * maxv + j)) * lane_stride;
            const u32 x0 = v[0], x1 = v[1 * lane_stride], x2 = v[2 * lane_stride], x3 = v[3 * lane_stride];
            c.vw0[s] = on ? x0 : 0u;
            c.vw1[s] = on ? x1 : 0u;
            a1lo[s] = on ? x2 : 0u;
            a1hi[s] = on ? x3 : 0u;
            types |= on ? 1u << ((x0 >> 24) & 0xFu) : 0u;
        }
    }
    {
        u64 syncs = 0;
        for (u32 d = 0; d < c.N; ++d) syncs |= (u64)(d + 1 < c.N ? (sel4(c.vw0, ord_slot(c, d + 1)) & 0xFFu) : c.L) << (8 * d);
        c.syncs = syncs;
        c.ends[0] = c.ends[1] = 0;
        u32 plain = 1;
#pragma unroll
        for (u32 s = 0; s < NS; ++s) {
            const u32 side = s / MV, j = s % MV;
            if (j < (side ? c.Q : c.T)) {
                c.ends[side] |= ((c.vw0[s] & 0xFFu) + ((c.vw0[s] >> 8) & 0xFFu)) << (10u * j);
                if ((c.vw1[s] & 0xFFu) == 0) plain = 0;
            }
        }
        c.plain = plain;
    }
    AVK_QF(1, c, 1)
    /* reference window: lane q writes the words k = q (mod 4) */
    if (!load_window<4>(a, c, h0, shift, q)) return AVK_LDEFER(4);
    AVK_QF(1, c, 2)
    qd_sync();
    /* FULL(side, mask): the 2 nm1 sequences dealt out over the four lanes, lengths and failed distances ORed together afterwards */
    {
        u32 job = 0;
        for (u32 m = 1; m <= c.nm1; ++m) {
            if (m < (1u << c.T)) {
                if ((job & 3u) == q) build_full<0>(c, m, a1lo, a1hi);
                job += 1;
            }
            if (m < (1u << c.Q)) {
                if ((job & 3u) == q) build_full<1>(c, m, a1lo, a1hi);
                job += 1;
            }
        }
        c.seq_len_lo = qd_or64(c.seq_len_lo);
        c.seq_len_hi = qd_or64(c.seq_len_hi);
        c.seq_fail_lo = qd_or64(c.seq_fail_lo);
        c.seq_fail_hi = qd_or64(c.seq_fail_hi);
    }
    qd_sync();
    AVK_QF(1, c, 3)
    AVK_LT_MARK(c, 0)

    /* ---- phase A: the lane's front is one of the four behind the queue and the kept states */
    const u32 off_dyn = c.off_q;
    c.off_wf = off_dyn + c.qcap + 2 * c.pool * c.wfr + q * c.wfr;
    u32 best_cost = 0;
    const int nopt = phaseA_quad(c, q, best_cost);
    AVK_LT_MARK(c, 1)
    if (nopt == LS_DEFER) return LS_DEFER;
    if (nopt < 0) return -nopt - 100;
    out.n_opt = (u32)nopt;
    if (a.mode == 1) { /* merge_solver.rs:137-143 */
        out.ed1 = best_cost == 0 ? 1u : 0u;
        return AVK_ST_OK;
    }
    qd_sync(); /* lane 0's optima list, the end of the shared queue and pool */

    /* ---- from here on DYN is four private arrays */
    const u32 k = q & 1u;
    const u32 prow = dyn_rows >> 2;
    const u32 qcap_search = c.qcap;
    c.off_wf = off_dyn + q * prow;
    c.off_q = c.off_wf;
    c.qcap = prow;
    c.wfcap_c = 4 * prow;
    if (max_ed_c && 2 * max_ed_c + 3 < c.wfcap_c) c.wfcap_c = 2 * max_ed_c + 3;

    /* ---- phase B for every tied optimum (waffle_solver.rs:169-261), one haplotype per lane; the first optimum with the fewest flips wins (:264-265) */
    u32 best_total = 0xFFFFFFFFu;
    Hap wh;
    hap_init(wh);
    u32 o_t = 0, o_q = 0;
    for (u32 kk = 0; kk < (u32)nopt; ++kk) {
        const u32 code = opt_get(c, kk);
        if (kk) { /* the mirror image of an earlier optimum cannot win and fails where the earlier one would have (avk_lane.inl) */
            const u32 mirror = ((code & 0x555u) << 1) | ((code >> 1) & 0x555u);
            bool seen = false;
            for (u32 j = 0; j < kk && mirror != code; ++j) seen = seen || opt_get(c, j) == mirror;
            if (seen) continue;
        }
        Hap h;
        bool dfr = false;
        if (best_cost == 0) { /* nothing skipped, no edits: the finished haplotypes are equal sequences */
            hapq_replay_zero(c, h, k, code, c.N);
            hap_step(c, h, true, false, 0, L_REF, c.L);
            h.d0 = h.t_len;
        } else {
            dfr = hapq_replay_exact(c, h, k, code, c.N);
            if (!dfr) {
                hap_step(c, h, true, false, 0, L_REF, c.L);
                int r = hap_update(c, h, 0);
                if (r == 0) r = hap_finalize(c, h, 0);
                dfr = r != 0;
            }
        }
        u32 rt = 0, rq = 0;
        const int e = dfr ? (int)LS_DEFER : gt_for_hap(c, h, rt, rq);
        const int eo = (int)qd_xor(1, (u32)e);
        const int e0 = k ? eo : e, e1 = k ? e : eo;
        if (e0 == LS_DEFER) return LS_DEFER;
        if (e0 < 0) return -e0 - 100;
        if (e1 == LS_DEFER) return LS_DEFER;
        if (e1 < 0) return -e1 - 100;
        const u32 total = (u32)e0 + (u32)e1;
        if (total < best_total) {
            best_total = total;
            wh = h;
            o_t = rt;
            o_q = rq;
            if (total == 0) break;
        }
    }
    /* the winner's two haplotypes on every lane */
    WinHap w0, w1;
    {
        const u32 wa = wh.t_alt | (wh.q_alt << 3) | (o_t << 6) | (o_q << 9) | (wh.ed << 12) | (wh.nskip << 20);
        const u32 wb = wh.t_skip | (wh.q_skip << 8) | (wh.t_len << 16) | (wh.q_len << 24);
        const u32 a0 = qd_bcast(0, wa), b0 = qd_bcast(0, wb), a1 = qd_bcast(1, wa), b1 = qd_bcast(1, wb);
        win_unpack(w0, a0, b0);
        win_unpack(w1, a1, b1);
    }
    out.ed1 = w0.ed;
    out.ed2 = w1.ed;
    AVK_LT_MARK(c, 2)

    /* ---- phase C: compare_expected_observed (:296-327) + per-call outputs (lane 0 writes) */
    u32 exp_pack = 0, obs_pack = 0;
    int bad = 0;
#pragma unroll
    for (u32 s = 0; s < NS; ++s) {
        const u32 side = s / MV, j = s % MV;
        const bool on = j < (side ? c.Q : c.T);
        const u32 b0 = ((side ? w0.q_alt : w0.t_alt) >> j) & 1u, b1 = ((side ? w1.q_alt : w1.t_alt) >> j) & 1u;
        const u32 o0 = ((side ? w0.o_q : w0.o_t) >> j) & 1u, o1 = ((side ? w1.o_q : w1.o_t) >> j) & 1u;
        const u32 ex = on ? b0 + b1 : 0u, ob = on ? o0 + o1 : 0u;
        exp_pack |= ex << (2 * s);
        obs_pack |= ob << (2 * s);
        if (!on) continue;
        if (ex == 0) bad = AVK_ST_VARIANT_METRICS;
        else if (ex < ob) bad = AVK_ST_TRUTH_FP;
        u32 cls = ex == ob ? AVK_CLASS_TP : AVK_CLASS_FN;
        u32 ea = ex, oa = ob;
        if (side) {
            if (cls == AVK_CLASS_FN) cls = AVK_CLASS_FP;
            ea = ob;
            oa = ex;
        }
        const u32 rz = b0 && b1 ? AVK_ZYG_HOM_ALT : (b0 ? AVK_ZYG_PHASED_HET10 : AVK_ZYG_PHASED_HET01);
        if (q == 0) a.var_out[v_off + (side ? c.T + j : j)] = ea | (oa << 8) | (cls << 16) | (rz << 24);
    }
    if (bad) return bad;

    /* add_basepair_stats (:335-449): lane (haplotype hh = q & 1, side sd = q >> 1) computes ed(ref, that side of that haplotype) where avk_lane.inl would */
    const u32 SUPMASK = (1u << AVK_VT_SNV) | (1u << AVK_VT_INSERTION) | (1u << AVK_VT_DELETION) | (1u << AVK_VT_INDEL) | (1u << AVK_VT_TR_CONTRACTION) |
                        (1u << AVK_VT_TR_EXPANSION) | (1u << AVK_VT_SV_DELETION) | (1u << AVK_VT_SV_INSERTION);
    out.present = types | SUPMASK;
    const bool same_haps = w0.t_alt == w1.t_alt && w0.q_alt == w1.q_alt;
    const u32 hh_me = q & 1u, sd_me = q >> 1;
    const WinHap &wme = hh_me ? w1 : w0;
    u32 X0, Y0, tp0, X1, Y1, tp1;
    {
        int er = 0;
        bool dfr = false;
        if (!(hh_me && same_haps)) {
            if (sd_me == 0) {
                if (wme.t_alt) er = ed_to_ref(c, 0, wme.t_alt, wme.t_len);
            } else if (wme.q_alt && !(wme.ed == 0 && wme.t_alt))
                er = ed_to_ref(c, 1, wme.q_alt, wme.q_len);
            dfr = er < 0;
        }
        if (qd_any(dfr)) return LS_DEFER;
        /* the four distances on every lane: [hh][side] */
        const u32 e00 = qd_bcast(0, (u32)er), e10 = qd_bcast(1, (u32)er), e01 = qd_bcast(2, (u32)er), e11 = qd_bcast(3, (u32)er);
        const u32 ert0 = e00, erq0 = w0.q_alt ? ((w0.ed == 0 && w0.t_alt) ? ert0 : e01) : (w0.ed == 0 ? ert0 : 0u);
        X0 = 2u * ert0, Y0 = 2u * erq0, tp0 = (X0 + Y0 - 2u * w0.ed) / 2u;
        if (same_haps) X1 = X0, Y1 = Y0, tp1 = tp0;
        else {
            const u32 ert1 = e10, erq1 = w1.q_alt ? ((w1.ed == 0 && w1.t_alt) ? ert1 : e11) : (w1.ed == 0 ? ert1 : 0u);
            X1 = 2u * ert1, Y1 = 2u * erq1, tp1 = (X1 + Y1 - 2u * w1.ed) / 2u;
        }
    }
    AVK_LT_MARK(c, 3)
    /* Alignments of the per-type groups (:383-445), BEFORE anything is added to the tally: entry (hh, side, first call of the type on the side) by lane
     * (hh, side); a second haplotype with the alleles of the first reads the first one's entries */
    {
        bool dfr = false;
        for (u32 left = types; left; left &= left - 1) {
            const u32 vt = (u32)__builtin_ctz(left);
            if (!((SUPMASK >> vt) & 1u)) continue;
            u32 tmask_g = 0, qmask_g = 0;
#pragma unroll
            for (u32 s = 0; s < NS; ++s) {
                const u32 side = s / MV, j = s % MV;
                if (j < (side ? c.Q : c.T) && ((c.vw0[s] >> 24) & 0xFu) == vt) (side ? qmask_g : tmask_g) |= 1u << j;
            }
            const u32 t_all = (1u << c.T) - 1u, q_all = (1u << c.Q) - 1u;
            const u32 side = sd_me, hh = hh_me;
            const u32 mask_g = side ? qmask_g : tmask_g;
            if (mask_g == 0 || mask_g == (side ? q_all : t_all)) continue; /* none of the type, or nothing but the type: no filtering */
            if (hh && same_haps) continue;
            const u32 e_idx = (hh * 2 + side) * MV + (u32)__builtin_ctz(mask_g);
            const u32 st = seq_id(c, 0, wme.t_alt), sq = seq_id(c, 1, wme.q_alt);
            const u32 Xh = hh ? X1 : X0, Yh = hh ? Y1 : Y0;
            const u32 alt = side ? wme.q_alt : wme.t_alt, m = alt & mask_g;
            u32 x2 = 0, z2 = (side ? Xh : Yh) / 2; /* nothing left of the side: it is the reference window */
            if (m && m == alt) { /* nothing filtered away on this haplotype: the side as it is */
                x2 = (side ? Yh : Xh) / 2;
                z2 = wme.ed;
            } else if (m) {
                const u32 sf = seq_id(c, side, m), fl = seq_len_of(c, sf);
                const int x = ed_to_ref(c, side, m, fl);
                const u32 gone = alt ^ m;
                int z = -1;
                if (wme.ed == 0 && wme.nskip == 0 && (gone & (gone - 1)) == 0 && seq_fail_of(c, side ? sq : st) == 0 && seq_fail_of(c, sf) == 0)
                    z = one_call_distance(c, MV * side + (u32)__builtin_ctz(gone));
                if (z < 0) z = side ? wfa_ed(c, st, wme.t_len, sf, fl) : wfa_ed(c, sf, fl, sq, wme.q_len);
                if (x < 0 || z < 0) {
                    dfr = true;
                    continue;
                }
                x2 = (u32)x;
                z2 = (u32)z;
            }
            filt_set(c, e_idx, x2 | (z2 << 8));
        }
        if (qd_any(dfr)) return LS_DEFER; /* (also: every lane's entries are written before any lane reads them) */
    }
    AVK_LT_MARK(c, 4)
    /* the groups: the joint one on every lane (its RECORD_BP check is the only way the region can still fail), the i-th group of the region added by lane i mod 4 */
    u32 *gm_out = a.group_metrics ? a.group_metrics + (u64)orig * AVK_N_GROUPS * AVK_N_FIELDS : (u32 *)0;
    if (gm_out) /* groups the region has nothing in: zeros (the others are written whole by the lane that adds them) */
        for (u32 g = 0; g < AVK_N_GROUPS; ++g)
            if (!(((1u | (types << 1)) >> g) & 1u))
                for (u32 i = q; i < AVK_N_FIELDS; i += 4) gm_out[g * AVK_N_FIELDS + i] = 0;
    u32 *bp_dst = a.bp_out ? a.bp_out + 4 * (u64)a.bp_off[orig] : (u32 *)0;
    qd_sync();
    u32 gi = 0;
    for (u32 left = 1u | (types << 1); left; left &= left - 1, ++gi) {
        const u32 g = (u32)__builtin_ctz(left);
        if (g != 0 && (gi & 3u) != q) continue;
        Group22 G;
#pragma unroll
        for (int i = 0; i < AVK_N_FIELDS; ++i) G.f[i] = 0;
        u32 tot_t = 0, tot_q = 0, tcount = 0, qcount = 0;
        u32 tmask_g = 0, qmask_g = 0;
#pragma unroll
        for (u32 s = 0; s < NS; ++s) {
            const bool on = (s % MV) < (s < MV ? c.T : c.Q);
            const u32 w0v = c.vw0[s], w1v = c.vw1[s];
            const u32 vt = (w0v >> 24) & 0xFu, z = (w0v >> 28) & 7u;
            if (!on || (g != 0 && vt != g - 1)) continue;
            if (s >= MV) g_add<true>(G, w1v & 0xFFu, (exp_pack >> (2 * s)) & 3u, (obs_pack >> (2 * s)) & 3u);
            else g_add<false>(G, w1v & 0xFFu, (exp_pack >> (2 * s)) & 3u, (obs_pack >> (2 * s)) & 3u);
            const u32 cntz = z == AVK_ZYG_HOM_ALT ? 2u : ((z == AVK_ZYG_UNPHASED_HET || z == AVK_ZYG_PHASED_HET01 || z == AVK_ZYG_PHASED_HET10) ? 1u : 0u);
            const u32 val = cntz * ((w1v >> 8) & 0xFFFFu);
            if (s < MV) {
                tot_t += val;
                tcount += 1;
                tmask_g |= 1u << s;
            } else {
                tot_q += val;
                qcount += 1;
                qmask_g |= 1u << (s - MV);
            }
        }
        if (g == 0) {
            G.f[AVK_F_BP_TRUTH_TP] += tp0 + tp1;
            G.f[AVK_F_BP_TRUTH_FN] += X0 - tp0 + 2 * w0.t_skip + X1 - tp1 + 2 * w1.t_skip; /* + skip metrics :378-381 */
            G.f[AVK_F_BP_QUERY_TP] += tp0 + tp1;
            G.f[AVK_F_BP_QUERY_FP] += Y0 - tp0 + 2 * w0.q_skip + Y1 - tp1 + 2 * w1.q_skip;
        } else if ((SUPMASK >> (g - 1)) & 1u) { /* :383-445 */
            for (u32 hh = 0; hh < 2; ++hh) {
                const WinHap &h = hh ? w1 : w0;
                const u32 Xh = hh ? X1 : X0, Yh = hh ? Y1 : Y0, tph = hh ? tp1 : tp0;
                const u32 hsrc = (hh && same_haps) ? 0u : hh; /* whose entries */
                u32 q_tp = 0, q_fp = 0, t_tp = 0, t_fn = 0;
                if (qcount) {
                    if (qcount == c.Q) {
                        q_tp = tph;
                        q_fp = Yh - tph + 2 * h.q_skip;
                    } else {
                        const u32 sf = seq_id(c, 1, h.q_alt & qmask_g);
                        const u32 e = filt_get(c, (hsrc * 2 + 1u) * MV + (u32)__builtin_ctz(qmask_g));
                        const u32 y2 = e & 0xFFu, z2 = e >> 8;
                        const u32 tp2 = (Xh + 2u * y2 - 2u * z2) / 2u;
                        q_tp = tp2;
                        q_fp = 2u * y2 - tp2 + 2 * seq_fail_of(c, sf);
                    }
                }
                if (tcount) {
                    if (tcount == c.T) {
                        t_tp = tph;
                        t_fn = Xh - tph + 2 * h.t_skip;
                    } else {
                        const u32 sf = seq_id(c, 0, h.t_alt & tmask_g);
                        const u32 e = filt_get(c, (hsrc * 2) * MV + (u32)__builtin_ctz(tmask_g));
                        const u32 x2 = e & 0xFFu, z2 = e >> 8;
                        const u32 tp2 = (2u * x2 + Yh - 2u * z2) / 2u;
                        t_tp = tp2;
                        t_fn = 2u * x2 - tp2 + 2 * seq_fail_of(c, sf);
                    }
                }
                G.f[AVK_F_BP_TRUTH_TP] += t_tp;
                G.f[AVK_F_BP_TRUTH_FN] += t_fn;
                G.f[AVK_F_BP_QUERY_TP] += q_tp;
                G.f[AVK_F_BP_QUERY_FP] += q_fp;
            }
        }
        /* add_record_basepair_stats (:455-522) */
        {
            const u32 tfn = G.f[AVK_F_BP_TRUTH_FN], qfp = G.f[AVK_F_BP_QUERY_FP];
            const u32 ttp = 2 * tot_t - tfn, qtp = 2 * tot_q - qfp;
            if (g == 0 && (ttp < G.f[AVK_F_BP_TRUTH_TP] || qtp < G.f[AVK_F_BP_QUERY_TP])) return AVK_ST_RECORD_BP;
            G.f[AVK_F_RBP_TRUTH_TP] += ttp;
            G.f[AVK_F_RBP_TRUTH_FN] += tfn;
            G.f[AVK_F_RBP_QUERY_TP] += qtp;
            G.f[AVK_F_RBP_QUERY_FP] += qfp;
        }
        if ((gi & 3u) != q) continue; /* (the joint group on the lanes that only checked it) */
#pragma unroll
        for (int i = 0; i < AVK_N_FIELDS; ++i) {
            const u32 v = G.f[i];
            if (gm_out) gm_out[g * AVK_N_FIELDS + i] = v;
            if (!v) continue;
            avk_tally_add_u32(tally, g * AVK_N_FIELDS + i, v);
        }
        if (bp_dst) {
            avk_u4 w;
            w.x = G.f[AVK_F_BP_TRUTH_TP], w.y = G.f[AVK_F_BP_TRUTH_FN], w.z = G.f[AVK_F_BP_QUERY_TP], w.w = G.f[AVK_F_BP_QUERY_FP];
            *(avk_u4 *)(bp_dst + 4 * gi) = w;
        }
    }
    (void)qcap_search;
    AVK_LT_MARK(c, 5)
    return AVK_ST_OK;
}

/* One persistent wave: claims 2^lanes_log2 (at most 16) fast records of a tile, every quad solves one.  wave_lds = this wave's rows,
 * wg_tally = the workgroup's 288 tally words in LDS (zeroed and flushed by the caller). */
#ifdef AVK_QUAD_WAVE_LOG
} // namespace quad
} // namespace avk
__device__ unsigned long long avk_wave_log_buf[6 * 4096 * 4]; /* [launch slot][wave][t0, t1, claims, longest claim] (s_memrealtime ticks) of the last step */
namespace avk {
namespace quad {
#endif
AVK_DEV void quad_worker(const AvkKernelArgs &a, const LaneArgs &la, u32 wave_id, u32 *wave_lds, u32 *wg_tally, u32 &n_ok_out, u32 &n_err_out, u64 *part = (u64 *)0) {
    const u32 lane = (u32)wv_lane();
    const u32 q = lane & 3u, rq = lane >> 2; /* lane of the quad, quad of the wave */
    const u32 width = 1u << la.lanes_log2, parts = 64u >> la.lanes_log2;
    LCtx c;
    c.ls = la.lanes_log2;
    c.p = wave_lds + (rq & (width - 1u));
    c.W1 = la.W + 1;
    c.nm1 = la.nm - 1;
    c.wfr = (2 * la.ed_max + 2 + 3) / 4;
    c.wfcap = 2 * la.ed_max + 2;
    c.optcap = lane_optcap(la.nm);
    c.off_opt = (1 + 2 * c.nm1) * c.W1;
    const u32 off_dyn = c.off_opt + c.optcap / 2;
    c.pool = la.pool < 8u ? la.pool : 8u;
    c.off_pool = off_dyn + la.qcap;
    const u32 dyn_rows = la.qcap + (2 * c.pool + 4) * c.wfr;
    c.max_nodes = la.max_nodes < 250u ? la.max_nodes : 250u;
    u32 n_ok = 0, n_err = 0, n_claims_done = 0;
    const u32 n_claims = la.n_tiles * parts;
#ifdef AVK_LANE_PHASE_TIMING
    for (int j = 0; j < 8; ++j) c.tph[j] = 0;
#endif
#ifdef AVK_QUAD_WAVE_LOG /* profiling build (make quad-wave-log, tools/gpu_quad_waves.py): when every wave of a quad launch started and ended, its claims, its longest claim */
    const u64 wl_t0 = __builtin_amdgcn_s_memrealtime();
    u64 wl_longest = 0, wl_prev = 0;
#endif
    for (;;) {
        u32 t = 0;
#ifdef AVK_LANE_PHASE_TIMING
        const u64 t_tile0 = avk_clock();
#endif
#ifdef AVK_QUAD_WAVE_LOG
        const u64 wl_c0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (lane == 0) t = avk_atomic_add_u32_global(la.tile_counter, 1u);
        t = wv_uni(wv_shfl(t, 0));
#ifdef AVK_QUAD_WAVE_LOG
        if (n_claims_done && wl_c0 - wl_prev > wl_longest) wl_longest = wl_c0 - wl_prev;
        wl_prev = wl_c0;
#endif
        if (t >= n_claims) break;
        n_claims_done += 1;
        if (part && (n_claims_done & 15u) == 0) { /* the 32-bit LDS tally moves on to the 64-bit partial tally every 16 claims */
            wv_sync();
            for (u32 i = lane; i < AVK_N_GROUPS * AVK_N_FIELDS; i += 64) {
                const u32 v = avk_wg_xchg(wg_tally + i, 0u);
                if (v) avk_atomic_add_u64_global(part + i, v);
            }
            wv_sync();
        }
        const u32 rl = (t & (parts - 1u)) * width + rq; /* this quad's record in the tile */
        t >>= 6u - la.lanes_log2;
        if (rq >= width) continue; /* the other quads only take part in the wave's claims and flushes */
        const u32 *rec = la.recs + (u64)t * la.rec_words * 64u + rl;
        const u32 h1 = rec[64];
        if (h1 == 0xFFFFFFFFu) continue; /* a slot of the class's last tile may have no region */
        const u32 orig = rec[3 * 64];
        const u32 v_off = rec[2 * 64];
        LaneOut out;
        out.ed1 = out.ed2 = out.n_opt = out.present = 0;
        /* (the per-region rows of the context; solve_quad moves them between the phases) */
        c.off_q = off_dyn;
        c.qcap = la.qcap;
        c.wfcap_c = 0;
#ifdef AVK_LANE_PHASE_TIMING
        c.tlast = avk_clock();
#endif
#ifdef AVK_QUAD_REGION_TICKS /* profiling build (tools/gpu_quad_regions.py): what every region costs its quad, by phase, in the unused last group of its metric block */
        u64 qr_ph0[6];
        for (int j = 0; j < 6; ++j) qr_ph0[j] = c.tph[j];
        const u64 qr_t0 = avk_clock();
#endif
        const int st = solve_quad(a, c, q, rec, 64u, dyn_rows, la.max_ed_c, out, wg_tally);
#ifdef AVK_QUAD_REGION_TICKS /* (the two distances of the region's record carry the tile's ticks / 16 and those of its phasing search instead: results of this build are not results) */
        if (st == AVK_ST_OK) {
            out.ed1 = (u32)((avk_clock() - qr_t0) >> 4) | 0x40000000u;
            out.ed2 = (u32)((c.tph[1] - qr_ph0[1]) >> 4) | (la.nm << 28);
        }
#endif
#ifdef AVK_LANE_PHASE_TIMING
        c.tph[6] += avk_clock() - t_tile0;
#endif
        if (q != 0) continue; /* lane 0 of the quad reports */
        if (st == LS_DEFER) { /* hand over to the wave-per-region kernels */
            const u32 slot_o = avk_atomic_add_u32_global(a.overflow_count, 1u);
            a.overflow_list[slot_o] = la.gen_base + (t * 64u + rl);
        } else {
            uint32_t w4[4];
            if (st == AVK_ST_OK) {
                w4[0] = 0;
                w4[1] = out.ed1;
                w4[2] = out.ed2;
                w4[3] = out.n_opt | (out.present << 16);
                n_ok += 1;
            } else {
                w4[0] = (u32)st;
                w4[1] = w4[2] = w4[3] = 0;
                n_err += 1;
                const u32 nvar = ((h1 >> 12) & 3u) + ((h1 >> 14) & 3u);
                for (u32 j = 0; j < nvar; ++j) a.var_out[v_off + j] = 0;
                if (a.group_metrics)
                    for (u32 i = 0; i < AVK_N_GROUPS * AVK_N_FIELDS; ++i) a.group_metrics[(u64)orig * AVK_N_GROUPS * AVK_N_FIELDS + i] = 0;
                if (a.bp_out)
                    for (u32 i = 4 * a.bp_off[orig]; i < 4 * a.bp_off[orig + 1]; ++i) a.bp_out[i] = 0;
            }
            avk_u4 *dst = (avk_u4 *)(a.region_out + 4 * (u64)orig);
            avk_u4 v;
            v.x = w4[0];
            v.y = w4[1];
            v.z = w4[2];
            v.w = w4[3];
            *dst = v;
        }
    }
#ifdef AVK_LANE_PHASE_TIMING
    if (q == 0 && rq < width && (la.nm > 4 || la.nm == 4)) { /* the launches that end a step: the three-call class, the head of the two-call class */
        u64 *pc = a.tally + (u64)(wave_id % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE + AVK_TALLY_LEN + 5 + (la.nm > 4 ? 0 : 8);
        for (int j = 0; j < 7; ++j) avk_atomic_add_u64_global(pc + j, c.tph[j]);
        avk_atomic_add_u64_global(pc + 7, 1);
    }
#else
    (void)wave_id;
#endif
#ifdef AVK_QUAD_WAVE_LOG
    if (lane == 0 && wave_id < 4096u) { /* launch slot: class (one, two, three calls per side) x width (16 or narrower) */
        unsigned long long *e = avk_wave_log_buf + ((size_t)((la.nm == 2 ? 0u : (la.nm == 4 ? 1u : 2u)) * 2u + (la.lanes_log2 == 4 ? 0u : 1u)) * 4096u + wave_id) * 4u;
        e[0] = wl_t0, e[1] = __builtin_amdgcn_s_memrealtime(), e[2] = n_claims_done, e[3] = wl_longest;
    }
#endif
    n_ok_out = n_ok;
    n_err_out = n_err;
}

} // namespace quad
} // namespace avk
#endif
