/*
 * avk_lane.inl — the compare solver for SMALL regions, one region per LANE (64 regions per wavefront).
 *
 * avk_solver.inl gives a whole 64-lane wavefront to one region; for the modal region of a call-set comparison (one or two
 * calls per side on a ~100-base window: 97 % of a whole-genome run) that leaves most lanes idle and spends ~30 bookkeeping
 * instructions per base compared.  Here every lane runs the SAME algorithm (reference solve_compare_region,
 * src/waffle_solver.rs:122-284, with optimize_sequences src/query_optimizer.rs:166-365, optimize_gt_alleles
 * src/exact_gt_optimizer.rs:108-357 and the dynamic wavefront aligner src/dwfa/dynamic_wfa.rs:23-276) on its own region:
 *
 *   - sequences are 2 bits per base, 16 bases per word; comparing two sequences is XOR + count-trailing-zeros on words;
 *   - a haplotype under construction is never copied: with at most MAXV calls per side there are at most 2^MAXV distinct
 *     full-length haplotype sequences per side; they are built once per region (FULL(side, mask), the rule of
 *     generate_allele_sequence, waffle_solver.rs:726-778, which is also the rule of HaplotypeTracker::extend_variant,
 *     haplotype_dwfa.rs:175-212, because a side's calls are sorted and sync points never pass a later call) and a search node's
 *     sequence is the PREFIX of FULL(side, alleles chosen so far) of the node's current length;
 *   - a search node is not stored either: a queue entry is ONE word (cost | node id | allele choices | depth) and the node's
 *     aligner state is rebuilt by replaying its at most 4 extension steps from the root when it is popped.  Pop order, node
 *     ids, the per-depth quota and the order of tied optima are exactly the reference's;
 *   - per-lane arrays (sequence table, wavefronts, queue) live in LDS as [word][lane]: bank = lane, conflict-free for any
 *     per-lane index.
 *
 * A lane that meets something outside this kernel's class (a non-ACGT base in its window, a wavefront or queue that would
 * outgrow the launch's capacities) appends its region to the overflow list of the wave-per-region kernels: results never
 * depend on which kernel solved a region (tests/test_emu_parity.py, tests/test_gpu_parity.py compare both against the oracle).
 */
#ifndef AVK_LANE_INL
#define AVK_LANE_INL

#include "avk_dev_types.h"
#include "avk_wave.h"

namespace avk {
namespace lane {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

/* work counters of instrumented emulator builds (-DAVK_LANE_STATS): [0] match_run words, [1] diagonals extended, [2] extension steps,
 * [3] pops, [4] partial re-pops, [5] regions, [6] phase-C alignments, [7] replayed steps */
#ifdef AVK_LANE_STATS
extern uint64_t g_lane_stats[32];
extern int g_lane_phase;
extern uint32_t *g_lane_work; /* per region (caller order): a weighted count of what its lane did (tools/lane_stats.py) */
extern uint32_t *g_lane_comp; /* per region: the first 16 counters above, as counted for that region alone ([13] steps of zero-cost replays, [14] pops of
                                 the genotype searches, [15] their replayed steps) */
static inline uint64_t lane_work_now() { return 30 * g_lane_stats[0] + 60 * g_lane_stats[1] + 150 * g_lane_stats[2] + 100 * g_lane_stats[3] + 300 * g_lane_stats[6] + 600 * g_lane_stats[5]; }
#define AVK_LSTAT(k, n) g_lane_stats[k] += (n)
#define AVK_LPHASE(k) g_lane_phase = (k)
#define AVK_LDEFER(k) (g_lane_stats[8 + (k)] += 1, g_lane_stats[16 + g_lane_phase] += ((k) == 0), (int)LS_DEFER)
#else
#define AVK_LSTAT(k, n)
#define AVK_LPHASE(k)
#define AVK_LDEFER(k) ((int)LS_DEFER)
#endif

/* profiling builds (-DAVK_LANE_PHASE_TIMING, make lane-timing): clock ticks per phase of solve_lane, summed per lane and added to the
 * spare words of the partial tally at the end of the launch (avk_debug_phase_cycles reads them after a download):
 * [0] record + reference window + sequence tables, [1] search A, [2] optimum replay + genotype search, [3] per-call outputs + edit
 * distances to the reference, [4] per-type alignments, [5] metric groups + tally, [6] whole tiles (claim to result), [7] lanes */
#ifdef AVK_LANE_PHASE_TIMING
#define AVK_LT_MARK(c, k)                     \
    {                                         \
        const u64 n_ = avk_clock();           \
        (c).tph[k] += n_ - (c).tlast;         \
        (c).tlast = n_;                       \
    }
#else
#define AVK_LT_MARK(c, k)
#endif

enum { L_REF = 1, L_ALT = 2 };
enum { LS_DEFER = -1 }; /* internal: not this kernel's class after all, hand the region to the wave-per-region kernels */

enum { MV = AVK_FAST_MAXV, NS = 2 * AVK_FAST_MAXV }; /* call slots per side, call slots of a region */

/* per-lane view of the wave's LDS: word (row, lane) sits at p[row << ls], ls = log2 of the lanes that work (the tile width of the launch) */
struct LCtx {
    u32 *p;
    u32 W1;      /* words per sequence (W + 1: the word after the last one is read by unaligned extracts) */
    u32 nm1;     /* masks per side minus one (1 or 3): sequence id = 0 reference, 1 + m - 1 truth mask m, 1 + nm1 + m - 1 query mask m */
    u32 off_wf;  /* first row of the wavefront byte arrays: hap 0, hap 1, scratch */
    u32 ls;      /* log2 of the LDS row stride in words (= lanes per tile) */
    u32 wfr;     /* rows per wavefront array */
    u32 wfcap_c; /* bytes of the metrics phase's one array (wfa_ed) */
#ifdef AVK_LANE_PHASE_TIMING
    mutable u64 tph[8], tlast;
#endif
#ifdef AVK_LANE_SLOW_TILES
    mutable u32 n_pops, n_diag, n_words; /* this lane's own loop turns in the current record (lanes share time, not turns) */
#define AVK_LCOUNT(c, f) (c).f += 1
#else
#define AVK_LCOUNT(c, f)
#endif
    u32 wfcap;   /* entries per wavefront array */
    u32 off_q, qcap;
    u32 off_opt, optcap;
    u32 off_pool, pool; /* kept node states of the search (NodePool): first row, slots */
    /* the region */
    u32 L, T, Q, N, ord; /* ord: 3 bits per search depth = slot of the call handled there */
    u32 takeq;           /* bit d: depth d handles a query call */
    u64 syncs;           /* byte d: the sync point behind depth d's step (the next call's position, the window's length behind the last) */
    u32 ends[2];         /* 10 bits per call of a side: where its REF allele ends in the window (position + REF length) */
    u32 plain;           /* 1: no call's alleles are at distance 0 from each other, so a node of cost 0 has skipped no call (nodeA_restore_zero) */
    u32 vw0[NS], vw1[NS]; /* slots [0, MV) truth, [MV, 2 MV) query: rel_pos | a0_len << 8 | a1_len << 16 | type << 24 | zyg << 28 ; alt_ed | raw_space << 8 */
    u64 seq_len_lo, seq_len_hi;   /* 8 bits per sequence id (ids 0-7, 8-15) */
    u64 seq_fail_lo, seq_fail_hi; /* failed_ed per sequence id (generate_allele_sequence, :745-753) */
    u32 max_branch;
    u32 max_nodes; /* a search that makes more nodes than this is handed over (LaneArgs::max_nodes) */
    /* The primitives below (sequence compares, the wavefront aligner, haplotype steps, the distances of the metrics phase) are written against these
     * accessors, so that another kernel can run them on a different layout: avk_wide.inl keeps one region's tables in a wave's LDS and lets every lane
     * work on a piece of that region. */
    enum { MV = AVK_FAST_MAXV };
    AVK_DEV_M u32 *seq_word(u32 s, u32 k) const { return p + ((s * W1 + k) << ls); } /* word k of sequence s */
    AVK_DEV_M u32 seq_stride() const { return 1u << ls; }                                   /* distance to the sequence's next word */
    AVK_DEV_M u32 seq_at(u32 k) const { return k << ls; }                                   /* word k of a sequence, from the sequence's word 0 */
    AVK_DEV_M u32 *wf_row(u32 arr, u32 row) const { return p + ((off_wf + arr * wfr + row) << ls); } /* entries 4 row .. 4 row + 3 of wavefront array arr */
    AVK_DEV_M u32 vw0_at(u32 slot) const;
    AVK_DEV_M u32 vw1_at(u32 slot) const;
    AVK_DEV_M u32 step_w0(u32 slot) const; /* vw0_at for the call a haplotype step applies (avk_wide.inl has that word at hand before it knows the slot) */
    AVK_DEV_M u32 vw0_side(u32 side, u32 j) const { return side ? vw0[AVK_FAST_MAXV + j] : vw0[j]; } /* j static: the records stay in registers */
    AVK_DEV_M u32 seq_id(u32 side, u32 mask) const { return mask == 0 ? 0u : 1u + side * nm1 + (mask - 1u); }
    AVK_DEV_M u32 seq_len(u32 s) const { return (u32)((s < 8 ? seq_len_lo : seq_len_hi) >> (8 * (s & 7u))) & 0xFFu; }
    AVK_DEV_M u32 seq_fail(u32 s) const { return (u32)((s < 8 ? seq_fail_lo : seq_fail_hi) >> (8 * (s & 7u))) & 0xFFu; }
};

AVK_DEV u32 v_pos(const LCtx &c, u32 s) { return c.vw0[s] & 0xFFu; }
AVK_DEV u32 v_a0(const LCtx &c, u32 s) { return (c.vw0[s] >> 8) & 0xFFu; }
AVK_DEV u32 v_a1(const LCtx &c, u32 s) { return (c.vw0[s] >> 16) & 0xFFu; }
AVK_DEV u32 v_type(const LCtx &c, u32 s) { return (c.vw0[s] >> 24) & 0xFu; }
AVK_DEV u32 v_zyg(const LCtx &c, u32 s) { return (c.vw0[s] >> 28) & 0x7u; }
AVK_DEV u32 v_alt_ed(const LCtx &c, u32 s) { return c.vw1[s] & 0xFFu; }
AVK_DEV u32 v_raw(const LCtx &c, u32 s) { return (c.vw1[s] >> 8) & 0xFFFFu; }
/* dynamic slot index: a select chain over the register-resident records */
AVK_DEV u32 sel4(const u32 (&a)[NS], u32 i) {
    const u32 v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = a[4], v5 = a[5]; /* every element read, selects on the values */
    u32 r = v0;
    r = i == 1 ? v1 : r;
    r = i == 2 ? v2 : r;
    r = i == 3 ? v3 : r;
    r = i == 4 ? v4 : r;
    r = i == 5 ? v5 : r;
    return r;
}

AVK_DEV_M u32 LCtx::vw0_at(u32 slot) const { return sel4(vw0, slot); }
AVK_DEV_M u32 LCtx::vw1_at(u32 slot) const { return sel4(vw1, slot); }
AVK_DEV_M u32 LCtx::step_w0(u32 slot) const { return sel4(vw0, slot); }
AVK_DEV u32 seq_id(const LCtx &c, u32 side, u32 mask) { return c.seq_id(side, mask); }
AVK_DEV u32 seq_len_of(const LCtx &c, u32 s) { return c.seq_len(s); }
AVK_DEV u32 seq_fail_of(const LCtx &c, u32 s) { return c.seq_fail(s); }

/* 16 bases of sequence s starting at base `off` (bits beyond the sequence's end are whatever the table holds) */
template <class C> AVK_DEV u32 extract16(const C &c, u32 s, u32 off) {
    const u32 k = off >> 4, sh = (off & 15u) * 2u;
    const u32 *w = c.seq_word(s, k);
    const u32 lo = w[0], hi = w[c.seq_stride()];
    return (u32)((((u64)hi << 32) | lo) >> sh);
}

/* number of positions on which a[ia..la) and b[ib..lb) agree before the first difference or either end */
template <class C> AVK_DEV u32 match_run(const C &c, u32 sa, u32 ia, u32 la, u32 sb, u32 ib, u32 lb) {
    const u32 ra = la > ia ? la - ia : 0u, rb = lb > ib ? lb - ib : 0u;
    const u32 lim = ra < rb ? ra : rb;
    u32 n = 0;
    while (n < lim) {
        AVK_LCOUNT(c, n_words);
        AVK_LSTAT(0, 1);
        AVK_LSTAT(24 + g_lane_phase, 1);
        const u32 x = extract16(c, sa, ia + n) ^ extract16(c, sb, ib + n);
        if (x) {
            n += (u32)__builtin_ctz(x) >> 1;
            break;
        }
        n += 16;
    }
    return n < lim ? n : lim;
}
/* The same for two sequences read from the SAME offset i (the zero-distance front of a haplotype, whose two strings are compared
 * position by position): whole table words, no unaligned extracts, and four words of each sequence per LDS round trip — a lane is a
 * chain of dependent LDS accesses, and this slide over the common part is most of what a cheap region does. */
template <class C> AVK_DEV u32 match_run_same(const C &c, u32 sa, u32 la, u32 sb, u32 lb, u32 i) {
#ifdef AVK_NO_MRS
    return match_run(c, sa, i, la, sb, i, lb);
#endif
    const u32 lmin = la < lb ? la : lb;
    if (i >= lmin) return 0;
    const u32 lim = lmin - i;
    const u32 *pa = c.seq_word(sa, 0), *pb = c.seq_word(sb, 0);
    u32 k = i >> 4;
    const u32 kend = (lmin + 15u) >> 4; /* words that hold bases below lmin */
    const u32 sh = (i & 15u) * 2u;
    u32 n = 0;
    { /* the word the run starts in */
        AVK_LSTAT(0, 1);
        AVK_LSTAT(24 + g_lane_phase, 1);
        const u32 x = (pa[c.seq_at(k)] ^ pb[c.seq_at(k)]) >> sh;
        if (x) {
            n = (u32)__builtin_ctz(x) >> 1;
            return n < lim ? n : lim;
        }
        n = 16u - (i & 15u);
        k += 1;
    }
    while (k < kend) { /* (words past kend are read from k: rows past the table's end are never touched) */
        const u32 k1 = k + 1 < kend ? k + 1 : k, k2 = k + 2 < kend ? k + 2 : k, k3 = k + 3 < kend ? k + 3 : k;
        const u32 a0 = pa[c.seq_at(k)], b0 = pb[c.seq_at(k)], a1 = pa[c.seq_at(k1)], b1 = pb[c.seq_at(k1)];
        const u32 a2 = pa[c.seq_at(k2)], b2 = pb[c.seq_at(k2)], a3 = pa[c.seq_at(k3)], b3 = pb[c.seq_at(k3)];
        AVK_LSTAT(0, 1);
        AVK_LSTAT(24 + g_lane_phase, 1);
        const u32 x0 = a0 ^ b0, x1 = a1 ^ b1, x2 = a2 ^ b2, x3 = a3 ^ b3;
        const u32 have = kend - k; /* words left, this one included */
        /* the first word that differs or ends the run */
        u32 adv = 64u;
        if (x3 || have <= 3u) adv = 48u + (x3 && have > 3u ? ((u32)__builtin_ctz(x3) >> 1) : 0u);
        if (x2 || have <= 2u) adv = 32u + (x2 && have > 2u ? ((u32)__builtin_ctz(x2) >> 1) : 0u);
        if (x1 || have <= 1u) adv = 16u + (x1 && have > 1u ? ((u32)__builtin_ctz(x1) >> 1) : 0u);
        if (x0) adv = (u32)__builtin_ctz(x0) >> 1;
        n += adv;
        if (adv < 64u) break;
        k += 4;
    }
    return n < lim ? n : lim;
}

/* ---- wavefront byte arrays ---------------------------------------------------------------- */
template <class C> AVK_DEV u8 *wf_ptr(const C &c, u32 arr, u32 i) { return (u8 *)c.wf_row(arr, i >> 2) + (i & 3u); }
template <class C> AVK_DEV u32 wf_get(const C &c, u32 arr, u32 i) { return *wf_ptr(c, arr, i); }
template <class C> AVK_DEV void wf_set(const C &c, u32 arr, u32 i, u32 v) { *wf_ptr(c, arr, i) = (u8)v; }
template <class C> AVK_DEV u32 *wf_row(const C &c, u32 arr, u32 row) { return c.wf_row(arr, row); } /* entries 4 row .. 4 row + 3 */

/* DWFALite on (B = sequence sb of length bl, O = sequence so of length ol); wf[i] = symbols of O consumed on diagonal i,
 * baseline offset = wf[i] + ed - i (dynamic_wfa.rs:114).
 * A lane's time is a chain of dependent LDS round trips, so the front is handled a ROW (four diagonals, one LDS word) at a time: one
 * read for the four offsets, the sixteen table reads of their first 16-base comparisons issued together, one write — instead of
 * a read, two dependent pairs of reads and a write per diagonal.  Diagonals that agree beyond 16 bases carry on one by one (rare
 * behind the first edit).  Each pass also says whether the front touches an end (update's stopping rule, :220-231: bit 0) or
 * holds a full diagonal (finalize's, :237-245: bit 1), so nothing is read again to decide how to go on. */
enum { DW_TOUCH = 1, DW_FULL = 2 };
#ifndef AVK_DW_BATCH
#define AVK_DW_BATCH 2u
#endif
/* the four diagonals 4 row .. 4 row + 3 of a front of `nd` diagonals at distance `ed`, offsets in `w` (one byte each): extended, flags added */
template <class C> AVK_DEV u32 dw_extend_row(const C &c, u32 w, u32 row, u32 nd, u32 ed, u32 sb, u32 bl, u32 so, u32 ol, u32 &flags) {
    u32 out = 0;
#pragma unroll
    for (u32 half = 0; half < 4; half += AVK_DW_BATCH) { /* AVK_DW_BATCH diagonals' reads in flight at a time (registers) */
        u32 d[AVK_DW_BATCH], bo[AVK_DW_BATCH], lim[AVK_DW_BATCH], xa[AVK_DW_BATCH], xb[AVK_DW_BATCH];
#pragma unroll
        for (u32 k = 0; k < AVK_DW_BATCH; ++k) {
            const u32 i = 4 * row + half + k;
            d[k] = (w >> (8 * (half + k))) & 0xFFu;
            bo[k] = d[k] + ed - i;
            const u32 ra = bl > bo[k] ? bl - bo[k] : 0u, rb = ol > d[k] ? ol - d[k] : 0u;
            lim[k] = i < nd ? (ra < rb ? ra : rb) : 0u;
        }
#pragma unroll
        for (u32 k = 0; k < AVK_DW_BATCH; ++k) { /* nothing to compare: read word 0 (offsets past an end may lie outside the table) */
            xa[k] = extract16(c, sb, lim[k] ? bo[k] : 0u);
            xb[k] = extract16(c, so, lim[k] ? d[k] : 0u);
        }
#pragma unroll
        for (u32 k = 0; k < AVK_DW_BATCH; ++k) {
            const u32 i = 4 * row + half + k;
            const u32 x = xa[k] ^ xb[k];
            u32 n = x ? (u32)__builtin_ctz(x) >> 1 : 16u;
            if (i < nd) {
                AVK_LSTAT(1, 1);
                AVK_LCOUNT(c, n_diag);
                AVK_LSTAT(0, lim[k] ? 1 : 0);
                AVK_LSTAT(24 + g_lane_phase, lim[k] ? 1 : 0);
            }
            if (n >= 16u && lim[k] > 16u) n = 16u + match_run(c, sb, bo[k] + 16u, bl, so, d[k] + 16u, ol);
            n = n < lim[k] ? n : lim[k];
            const u32 dn = d[k] + n;
            if (i < nd) {
                const bool eb = dn + ed - i >= bl, eo = dn >= ol;
                flags |= (eb || eo) ? (u32)DW_TOUCH : 0u;
                flags |= (eb && eo) ? (u32)DW_FULL : 0u;
            }
            out |= dn << (8 * (half + k));
        }
    }
    return out;
}
/* extend (:94-130): every diagonal as far as it matches */
template <class C> AVK_DEV u32 dw_extend(const C &c, u32 arr, u32 ed, u32 sb, u32 bl, u32 so, u32 ol) {
    const u32 nd = 2 * ed + 1;
    u32 flags = 0;
    for (u32 row = 0; 4 * row < nd; ++row) {
        u32 *pw = wf_row(c, arr, row);
        *pw = dw_extend_row(c, *pw, row, nd, ed, sb, bl, so, ol, flags);
    }
    return flags;
}
/* increase_edit_distance (:140-173: new[k] = max(old[k], old[k-1] + 1, old[k-2] + 1), no clipping) and the extend that always follows it,
 * in one pass over the rows, top row first (row r of the new front needs rows r and r - 1 of the old one) */
template <class C> AVK_DEV u32 dw_bump_extend(const C &c, u32 arr, u32 old_ed, u32 sb, u32 bl, u32 so, u32 ol) {
    const u32 nd = 2 * old_ed + 1, nn = nd + 2, ed = old_ed + 1;
    u32 flags = 0;
    int row = (int)((nn - 1) >> 2);
    auto old_row = [&](int r) -> u32 { /* entries >= nd of the old front do not exist */
        if (r < 0 || 4u * (u32)r >= nd) return 0u;
        const u32 w = *wf_row(c, arr, (u32)r);
        const u32 have = nd - 4u * (u32)r;
        return have >= 4u ? w : (w & ((1u << (8u * have)) - 1u));
    };
    u32 cur = old_row(row);
    for (; row >= 0; --row) {
        const u32 prev = old_row(row - 1);
        const u64 both = ((u64)cur << 32) | prev;
        const u32 b1 = (u32)(both >> 24), b2 = (u32)(both >> 16); /* entries k - 1 and k - 2 of the old front, byte k */
        u32 w = 0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const u32 g = 4u * (u32)row + k;
            u32 v = (cur >> (8 * k)) & 0xFFu; /* 0 where the old front has no entry */
            const u32 t1 = ((b1 >> (8 * k)) & 0xFFu) + 1u, t2 = ((b2 >> (8 * k)) & 0xFFu) + 1u;
            if (g >= 1u && g - 1u < nd) v = t1 > v ? t1 : v;
            if (g >= 2u && g - 2u < nd) v = t2 > v ? t2 : v;
            w |= (g < nn ? v : 0u) << (8 * k);
        }
        *wf_row(c, arr, (u32)row) = dw_extend_row(c, w, (u32)row, nn, ed, sb, bl, so, ol, flags);
        cur = prev;
    }
    return flags;
}
enum { LS_PARTIAL = 1 }; /* a capped alignment stopped: the distance is MORE than the budget (how much more is not known) */
/* update (:68-84): extend, then raise the distance until EITHER end is touched; LS_DEFER when the array is too small.
 * `budget` = the largest distance the caller cares about: LS_PARTIAL as soon as the distance is known to exceed it. */
template <class C> AVK_DEV int dw_update(const C &c, u32 arr, u32 &ed, u32 sb, u32 bl, u32 so, u32 ol, u32 budget) {
    u32 fl = dw_extend(c, arr, ed, sb, bl, so, ol);
    while (!(fl & DW_TOUCH)) {
        if (ed + 1 > budget) return LS_PARTIAL;
        if (2 * ed + 3 > c.wfcap) return AVK_LDEFER(0);
        fl = dw_bump_extend(c, arr, ed, sb, bl, so, ol);
        ed += 1;
    }
    return 0;
}
template <class C> AVK_DEV int dw_finalize(const C &c, u32 arr, u32 &ed, u32 sb, u32 bl, u32 so, u32 ol, u32 budget, u32 cap) { /* :183-198 */
    u32 fl = dw_extend(c, arr, ed, sb, bl, so, ol);
    while (!(fl & DW_FULL)) {
        if (ed + 1 > budget) return LS_PARTIAL;
        if (2 * ed + 3 > cap) return AVK_LDEFER(0);
        fl = dw_bump_extend(c, arr, ed, sb, bl, so, ol);
        ed += 1;
    }
    return 0;
}
/* wfa_ed (src/util/sequence_alignment.rs:9-13) = unit-cost edit distance of two complete sequences.  Only the metrics phase aligns this
 * way, when the search is over: the rows of the three wavefront arrays and of the queue are one long array for it (wfcap_c bytes). */
template <class C> AVK_DEV int wfa_ed(const C &c, u32 sa, u32 la, u32 sb, u32 lb) {
    AVK_LSTAT(6, 1);
    const u32 lim = la < lb ? la : lb;
    const u32 d = match_run_same(c, sa, la, sb, lb, 0);
    if (d == lim) return (int)((la > lb ? la : lb) - lim);
    wf_set(c, 0, 0, d);
    u32 ed = 0;
    if (dw_finalize(c, 0, ed, sa, la, sb, lb, 0xFFFFu, c.wfcap_c)) return LS_DEFER;
    return (int)ed;
}

/* ---- one haplotype of a search node (HaplotypeDWFA, haplotype_dwfa.rs:17-24) ----------------------------------------- */
struct Hap {
    u32 t_refpos, q_refpos, t_len, q_len, t_skip, q_skip, nskip;
    u32 t_alt, q_alt; /* alleles chosen so far: bit j = ALT for the side's call j */
    u32 t_nal, q_nal;
    u32 ed, d0;       /* while ed == 0 the wavefront is the single offset d0 */
};
AVK_DEV void hap_init(Hap &h) {
    h.t_refpos = h.q_refpos = h.t_len = h.q_len = h.t_skip = h.q_skip = h.nskip = 0;
    h.t_alt = h.q_alt = h.t_nal = h.q_nal = h.ed = h.d0 = 0;
}
/* HaplotypeDWFA::extend_variant without the aligner update (haplotype_dwfa.rs:46-62, :175-227): lengths and positions only,
 * the bases are implied by FULL(side, chosen alleles).  has_var false = the two copy_reference(region end) of finalize_dwfa. */
template <class C> AVK_DEV bool hap_step(const C &c, Hap &h, bool is_truth, bool has_var, u32 slot, u32 allele, u32 sync) {
    /* "this" side and the "other" side by value (selects), written back at the end: no addresses into the record are taken */
    u32 tl = is_truth ? h.t_len : h.q_len, ol = is_truth ? h.q_len : h.t_len;
    u32 trp = is_truth ? h.t_refpos : h.q_refpos, orp = is_truth ? h.q_refpos : h.t_refpos;
    u32 tskip = is_truth ? h.t_skip : h.q_skip, tnal = is_truth ? h.t_nal : h.q_nal, talt = is_truth ? h.t_alt : h.q_alt;
    const u32 n_o = orp < sync ? sync - orp : 0u;
    u32 n1 = 0, n2 = 0, rp = trp;
    bool ok = true;
    if (has_var) {
        const u32 w0 = c.step_w0(slot);
        const u32 pos = w0 & 0xFFu, a0 = (w0 >> 8) & 0xFFu, a1 = (w0 >> 16) & 0xFFu;
        if (rp < pos) {
            n1 = pos - rp;
            rp = pos;
        }
        if (allele == L_ALT) {
            if (rp <= pos) {
                n2 = a1;
                rp = pos + a0;
            } else {
                tskip += c.vw1_at(slot) & 0xFFu; /* edit_distance(allele0, allele1), :199 */
                h.nskip += 1;
                ok = false;
            }
            talt |= 1u << tnal;
        }
        tnal += 1;
    }
    const u32 n3 = rp < sync ? sync - rp : 0u;
    if (rp < sync) rp = sync;
    ol += n_o;
    if (n_o) orp = sync;
    tl += n1 + n2 + n3;
    trp = rp;
    h.t_len = is_truth ? tl : ol;
    h.q_len = is_truth ? ol : tl;
    h.t_refpos = is_truth ? trp : orp;
    h.q_refpos = is_truth ? orp : trp;
    h.t_skip = is_truth ? tskip : h.t_skip;
    h.q_skip = is_truth ? h.q_skip : tskip;
    h.t_nal = is_truth ? tnal : h.t_nal;
    h.q_nal = is_truth ? h.q_nal : tnal;
    h.t_alt = is_truth ? talt : h.t_alt;
    h.q_alt = is_truth ? h.q_alt : talt;
    return ok;
}
/* DWFALite::update on the haplotype's two sequences (hap_update of avk_solver.inl) */
template <class C> AVK_DEV int hap_update(const C &c, Hap &h, u32 arr, u32 budget = 0xFFFFu) {
    const u32 st = c.seq_id(0, h.t_alt), sq = c.seq_id(1, h.q_alt);
    if (h.ed == 0) {
        h.d0 += match_run_same(c, st, h.t_len, sq, h.q_len, h.d0);
        const u32 lim = h.t_len < h.q_len ? h.t_len : h.q_len;
        if (h.d0 >= lim) return 0;
        if (budget == 0) return LS_PARTIAL;
        if (c.wfcap < 3) return AVK_LDEFER(0);
        wf_set(c, arr, 0, h.d0);
    }
    return dw_update(c, arr, h.ed, st, h.t_len, sq, h.q_len, budget);
}
template <class C> AVK_DEV int hap_finalize(const C &c, Hap &h, u32 arr, u32 budget = 0xFFFFu) {
    const u32 st = c.seq_id(0, h.t_alt), sq = c.seq_id(1, h.q_alt);
    if (h.ed == 0) {
        if (h.d0 >= h.t_len && h.d0 >= h.q_len) return 0;
        if (budget == 0) return LS_PARTIAL;
        wf_set(c, arr, 0, h.d0);
    }
    return dw_finalize(c, arr, h.ed, st, h.t_len, sq, h.q_len, budget, c.wfcap);
}
/* hap_update (final = false) or hap_update + hap_finalize (final = true: the second carries on where the first stops, and a front with a full diagonal touches
 * an end, so the two are ONE walk up the distances that ends at the later condition) — one copy of the aligner's loops where a kernel wants one call site */
template <class C> AVK_DEV int hap_align(const C &c, Hap &h, u32 arr, u32 budget, bool final) {
    const u32 st = c.seq_id(0, h.t_alt), sq = c.seq_id(1, h.q_alt);
    if (h.ed == 0) {
        h.d0 += match_run_same(c, st, h.t_len, sq, h.q_len, h.d0);
        const u32 lim = h.t_len < h.q_len ? h.t_len : h.q_len;
        if (final ? (h.d0 >= h.t_len && h.d0 >= h.q_len) : (h.d0 >= lim)) return 0;
        if (budget == 0) return LS_PARTIAL;
        if (c.wfcap < 3) return AVK_LDEFER(0);
        wf_set(c, arr, 0, h.d0);
    }
    const u32 want = final ? (u32)DW_FULL : (u32)DW_TOUCH;
    u32 fl = dw_extend(c, arr, h.ed, st, h.t_len, sq, h.q_len);
    while (!(fl & want)) {
        if (h.ed + 1 > budget) return LS_PARTIAL;
        if (2 * h.ed + 3 > c.wfcap) return AVK_LDEFER(0);
        fl = dw_bump_extend(c, arr, h.ed, st, h.t_len, sq, h.q_len);
        h.ed += 1;
    }
    return 0;
}
/* the search order (order_variants, query_optimizer.rs:372-381): slot of the call at depth d, and the step's sync point */
AVK_DEV u32 ord_slot(const LCtx &c, u32 d) { return (c.ord >> (3 * d)) & 7u; }
AVK_DEV u32 sync_after(const LCtx &c, u32 d) { return (u32)(c.syncs >> (8 * d)) & 0xFFu; }

/* ---- phase A: optimize_sequences ------------------------------------------------------------------------------------- */
/* a node = two haplotypes; code: 2 bits per depth, bit 0 = ALT on haplotype 1, bit 1 = ALT on haplotype 2 */
struct NodeA {
    Hap h[2];
};
AVK_DEV u32 nodeA_cost(const NodeA &n) { return n.h[0].t_skip + n.h[0].q_skip + n.h[0].ed + n.h[1].t_skip + n.h[1].q_skip + n.h[1].ed; }
/* One extension step of both haplotypes.  `cap` = the largest total node cost the caller cares about: the search only needs a
 * node's exact cost when the node can still be the next pop, and most wrongly phased branches never are — their alignments stop at
 * the first edit that proves cost > cap instead of running to a distance of tens (returns LS_PARTIAL and a lower bound > cap). */
/* the alignments of a step whose haplotype steps are made (also: of a kept node whose last step was stopped early, taken further) */
AVK_DEV int nodeA_settle(const LCtx &c, NodeA &n, u32 cap, u32 &lb) {
    const u32 base = n.h[0].t_skip + n.h[0].q_skip + n.h[1].t_skip + n.h[1].q_skip;
    lb = base + n.h[0].ed + n.h[1].ed; /* distances never decrease */
    if (lb > cap) return LS_PARTIAL;
    const u32 b0 = cap - base - n.h[1].ed;
    int r = hap_update(c, n.h[0], 0, b0);
    if (r == LS_PARTIAL) lb = cap + 1;
    if (r) return r;
    const u32 b1 = cap - base - n.h[0].ed;
    r = hap_update(c, n.h[1], 1, b1);
    if (r == LS_PARTIAL) lb = cap + 1;
    return r;
}
AVK_DEV int nodeA_step(const LCtx &c, NodeA &n, u32 d, u32 choice, u32 cap, u32 &lb) {
    const u32 slot = ord_slot(c, d);
    const bool is_truth = slot < MV;
    const u32 sync = sync_after(c, d);
    AVK_LSTAT(2, 1);
    hap_step(c, n.h[0], is_truth, true, slot, (choice & 1u) ? L_ALT : L_REF, sync);
    hap_step(c, n.h[1], is_truth, true, slot, (choice & 2u) ? L_ALT : L_REF, sync);
    return nodeA_settle(c, n, cap, lb);
}
AVK_DEV int nodeA_step(const LCtx &c, NodeA &n, u32 d, u32 choice) {
    u32 lb;
    return nodeA_step(c, n, d, choice, 0xFFFFu, lb);
}
AVK_DEV int nodeA_replay(const LCtx &c, NodeA &n, u32 code, u32 depth) {
    hap_init(n.h[0]);
    hap_init(n.h[1]);
    AVK_LSTAT(7, depth);
    for (u32 d = 0; d < depth; ++d)
        if (nodeA_step(c, n, d, (code >> (2 * d)) & 3u)) return LS_DEFER;
    return 0;
}
/* The same for a node whose exact cost is known to be 0: no call was skipped and neither haplotype has an edit, on the node and (costs
 * never decrease along a path) on all its ancestors.  Every update stopped at the end of the shorter sequence without a mismatch, so
 * the wavefront is that offset and nothing needs to be compared again. */
AVK_DEV void nodeA_replay_steps(const LCtx &c, NodeA &n, u32 code, u32 depth) { /* lengths, positions, skipped calls: everything but the alignments */
    hap_init(n.h[0]);
    hap_init(n.h[1]);
    AVK_LSTAT(13, depth);
    for (u32 d = 0; d < depth; ++d) {
        const u32 slot = ord_slot(c, d), choice = (code >> (2 * d)) & 3u, sync = sync_after(c, d);
        hap_step(c, n.h[0], slot < MV, true, slot, (choice & 1u) ? L_ALT : L_REF, sync);
        hap_step(c, n.h[1], slot < MV, true, slot, (choice & 2u) ? L_ALT : L_REF, sync);
    }
}
/* The same in closed form, for regions without calls whose alleles are at distance 0 (c.plain): a node of cost 0 has then skipped no call, so every ALT it chose
 * was spliced in where it stands — a side's reference position is the later of the last sync point and the end of the REF span of the side's last ALT call, its
 * length that position plus what the chosen ALTs add (the length of FULL(side, chosen) minus the window's).  A dozen operations per haplotype and side whatever
 * the depth, against a haplotype step per depth. */
AVK_DEV void nodeA_restore_zero(const LCtx &c, NodeA &n, u32 code, u32 depth) {
    u32 alt[2][2] = {{0, 0}, {0, 0}}; /* [haplotype][side] */
    u32 cnt[2] = {0, 0};
    for (u32 d = 0; d < depth; ++d) {
        const u32 ch = (code >> (2 * d)) & 3u, q = (c.takeq >> d) & 1u;
        const u32 at = q ? cnt[1] : cnt[0];
        const u32 b0 = (ch & 1u) << at, b1 = (ch >> 1) << at;
        alt[0][0] |= q ? 0u : b0, alt[0][1] |= q ? b0 : 0u;
        alt[1][0] |= q ? 0u : b1, alt[1][1] |= q ? b1 : 0u;
        cnt[0] += 1u - q, cnt[1] += q;
    }
    const u32 sync = depth ? sync_after(c, depth - 1) : 0u;
#pragma unroll
    for (u32 k = 0; k < 2; ++k) {
        Hap &h = n.h[k];
        u32 pos[2], len[2];
#pragma unroll
        for (u32 side = 0; side < 2; ++side) {
            const u32 m = alt[k][side];
            const u32 last = m ? (c.ends[side] >> (10u * (31u - (u32)__builtin_clz(m)))) & 0x3FFu : 0u;
            pos[side] = last > sync ? last : sync;
            len[side] = pos[side] + c.seq_len(c.seq_id(side, m)) - c.L;
        }
        h.t_refpos = pos[0], h.q_refpos = pos[1], h.t_len = len[0], h.q_len = len[1];
        h.t_skip = h.q_skip = h.nskip = 0;
        h.t_alt = alt[k][0], h.q_alt = alt[k][1], h.t_nal = cnt[0], h.q_nal = cnt[1];
        h.ed = 0;
        h.d0 = len[0] < len[1] ? len[0] : len[1];
    }
}
AVK_DEV void nodeA_replay_zero(const LCtx &c, NodeA &n, u32 code, u32 depth) {
    if (c.plain) {
        nodeA_restore_zero(c, n, code, depth);
        return;
    }
    nodeA_replay_steps(c, n, code, depth);
    n.h[0].d0 = n.h[0].t_len < n.h[0].q_len ? n.h[0].t_len : n.h[0].q_len;
    n.h[1].d0 = n.h[1].t_len < n.h[1].q_len ? n.h[1].t_len : n.h[1].q_len;
}
/* ---- kept node states.  A node is queued as its path (code, depth); what the path does not say is where its two alignments stand: the distance and the
 * front of each haplotype (while the distance is 0: the single offset d0).  Replaying the path to get them back is an alignment of the whole prefix per
 * pop — for the regions that make a launch long, most of the work.  So the fronts of queued nodes that HAVE a distance (or whose last step was stopped
 * early) are kept in a small pool of slots in the lane's rows, found again by the node's id; a node that finds no free slot is replayed as before.
 * Slot: 2 x wfr rows (the two haplotypes' front arrays as they are); ids and distances of the slots in two registers (a byte per slot). */
struct NodePool {
    u64 ids; /* byte s: id of the node in slot s, 0xFF = free (ids stay below 251) */
    u64 eds; /* byte s: ed of haplotype 0 | ed of haplotype 1 << 4 */
};
AVK_DEV u32 pool_find(const LCtx &c, const NodePool &pl, u32 id) {
    u32 hit = c.pool;
    for (u32 s = 0; s < c.pool; ++s)
        if (((u32)(pl.ids >> (8 * s)) & 0xFFu) == id) hit = s;
    return hit;
}
AVK_DEV void pool_free(NodePool &pl, u32 s) { pl.ids |= 0xFFull << (8 * s); }
AVK_DEV void pool_save(const LCtx &c, NodePool &pl, u32 s, u32 id, const NodeA &n) {
    pl.ids = (pl.ids & ~(0xFFull << (8 * s))) | ((u64)id << (8 * s));
    pl.eds = (pl.eds & ~(0xFFull << (8 * s))) | ((u64)(n.h[0].ed | (n.h[1].ed << 4)) << (8 * s));
    for (u32 k = 0; k < 2; ++k) {
        u32 *dst = c.p + ((c.off_pool + (2 * s + k) * c.wfr) << c.ls);
        if (n.h[k].ed == 0) {
            dst[0] = n.h[k].d0;
            continue;
        }
        const u32 rows = (2 * n.h[k].ed + 1 + 3) >> 2;
        for (u32 r = 0; r < rows; ++r) dst[r << c.ls] = *c.wf_row(k, r);
    }
}
AVK_DEV void pool_load(const LCtx &c, NodePool &pl, u32 s, NodeA &n) { /* the haplotype steps of n are made (nodeA_replay_steps); frees the slot */
    const u32 e = (u32)(pl.eds >> (8 * s)) & 0xFFu;
    n.h[0].ed = e & 15u, n.h[1].ed = e >> 4;
    for (u32 k = 0; k < 2; ++k) {
        const u32 *src = c.p + ((c.off_pool + (2 * s + k) * c.wfr) << c.ls);
        if (n.h[k].ed == 0) {
            n.h[k].d0 = src[0] & 0xFFu;
            continue;
        }
        const u32 rows = (2 * n.h[k].ed + 1 + 3) >> 2;
        for (u32 r = 0; r < rows; ++r) *c.wf_row(k, r) = src[r << c.ls];
    }
    pool_free(pl, s);
}
/* a queued node's state goes into a free slot, if there is one and the path alone does not say it all */
AVK_DEV void pool_keep(const LCtx &c, NodePool &pl, u32 id, const NodeA &n, bool partial) {
    if (!c.pool || !(partial || (n.h[0].ed | n.h[1].ed))) return;
    u32 s = c.pool;
    for (u32 k = 0; k < c.pool; ++k)
        if (((u32)(pl.ids >> (8 * k)) & 0xFFu) == 0xFFu) s = k;
    if (s >= c.pool) {
        AVK_LSTAT(22, 1);
        return;
    }
    pool_save(c, pl, s, id, n);
}
/* ComparisonNode::finalize_dwfas (:457-462, haplotype_dwfa.rs:84-95); LS_PARTIAL: the final cost is more than `cap` */
AVK_DEV int nodeA_finalize(const LCtx &c, NodeA &n, u32 cap = 0xFFFFu) {
    hap_step(c, n.h[0], true, false, 0, L_REF, c.L);
    hap_step(c, n.h[1], true, false, 0, L_REF, c.L);
    const u32 base = n.h[0].t_skip + n.h[0].q_skip + n.h[1].t_skip + n.h[1].q_skip;
    if (base + n.h[0].ed + n.h[1].ed > cap) return LS_PARTIAL;
    int r = hap_update(c, n.h[0], 0, cap - base - n.h[1].ed);
    if (r) return r;
    r = hap_finalize(c, n.h[0], 0, cap - base - n.h[1].ed);
    if (r) return r;
    r = hap_update(c, n.h[1], 1, cap - base - n.h[0].ed);
    if (r) return r;
    return hap_finalize(c, n.h[1], 1, cap - base - n.h[0].ed);
}
/* pops the smallest queue word (the entries of both searches are built so that this is the reference's next pop) */
AVK_DEV u32 q_pop_min(const LCtx &c, u32 &qn) {
    u32 best = 0xFFFFFFFFu, bi = 0;
    for (u32 i = 0; i < qn; ++i) {
        const u32 e = c.p[(c.off_q + i) << c.ls];
        if (e < best) {
            best = e;
            bi = i;
        }
    }
    qn -= 1;
    c.p[(c.off_q + bi) << c.ls] = c.p[(c.off_q + qn) << c.ls];
    return best;
}
AVK_DEV int q_push(const LCtx &c, u32 &qn, u32 e) {
    if (qn >= c.qcap) return AVK_LDEFER(1);
    c.p[(c.off_q + qn) << c.ls] = e;
    qn += 1;
    return 0;
}
/* after the searches the same rows hold the (x, z) pairs of the per-type alignments, 16 bits each */
AVK_DEV u32 filt_get(const LCtx &c, u32 k) { return *((uint16_t *)(c.p + ((c.off_opt + (k >> 1)) << c.ls)) + (k & 1u)); }
AVK_DEV void filt_set(const LCtx &c, u32 k, u32 v) { *((uint16_t *)(c.p + ((c.off_opt + (k >> 1)) << c.ls)) + (k & 1u)) = (uint16_t)v; }
AVK_DEV u32 opt_get(const LCtx &c, u32 k) { return *((uint16_t *)(c.p + ((c.off_opt + (k >> 1)) << c.ls)) + (k & 1u)); }
AVK_DEV void opt_set(const LCtx &c, u32 k, u32 v) { *((uint16_t *)(c.p + ((c.off_opt + (k >> 1)) << c.ls)) + (k & 1u)) = (uint16_t)v; }

/* Queue words of phase A: cost << 24 | id << 16 | code << 4 | partial << 3 | depth (code: 2 bits per depth, up to 6 depths; a cost
 * beyond 255 hands the region over).  A `partial` entry carries a LOWER BOUND of the
 * node's cost (its last step was stopped early, nodeA_step).  The reference pops the minimum of (cost, id); here the minimum word is
 * popped, and when it is partial its cost is worked out further — up to the next entry's cost, which is as far as it can matter —
 * and it goes back into the queue unless it turns out to cost exactly what it was popped for.  Lower bounds never exceed the true
 * cost, ids are unique, so the sequence of REAL pops (and with it ids, quota counts and the order of tied optima) is the reference's. */
AVK_DEV u32 q_min_cost(const LCtx &c, u32 qn) {
    u32 best = 0xFFFFFFFFu;
    for (u32 i = 0; i < qn; ++i) {
        const u32 e = c.p[(c.off_q + i) << c.ls];
        best = e < best ? e : best;
    }
    return qn ? best >> 24 : 0xFFFFu;
}
AVK_DEV u32 keyA(u32 cost, u32 id, u32 code, u32 partial, u32 depth) { return (cost << 24) | (id << 16) | (code << 4) | (partial << 3) | depth; }
AVK_DEV int pushA(const LCtx &c, u32 &qn, u32 cost, u32 id, u32 code, u32 partial, u32 depth) {
    if (cost > 255u) return AVK_LDEFER(5);
    return q_push(c, qn, keyA(cost, id, code, partial, depth));
}

/* returns the number of tied optima (codes in the opt array, in the order the reference finds them), LS_DEFER, or -100 - status */
AVK_DEV int phaseA(const LCtx &c, u32 &best_out) {
    u32 qn = 0;
    q_push(c, qn, 0);
    u32 next_id = 1, best = 0xFFFFu, nbest = 0;
    u64 bucket = 0; /* 8 bits per depth (fewer than 250 nodes are ever made) */
    NodePool pl;
    pl.ids = ~0ull, pl.eds = 0;
    while (qn > 0) {
        const u32 e = q_pop_min(c, qn);
        const u32 cost = e >> 24;
        if (cost > best) break; /* :204; pops come in non-decreasing cost order */
        AVK_LSTAT(3, 1);
        AVK_LCOUNT(c, n_pops);
        AVK_LSTAT(4, (e >> 3) & 1u);
        const u32 depth = e & 7u, code = (e >> 4) & 0xFFFu, id = (e >> 16) & 0xFFu;
        NodeA n;
        bool have = false; /* n holds the node's state */
        if (e & 8u) { /* partial: ancestors are exact (they were popped), the last step is taken further */
            /* as far as it can matter for the order (the next entry's cost), and at least doubling, so that a node that really is
             * expensive is taken up a logarithmic number of times; never beyond the best finished cost (:204 drops it there) */
            u32 cap = q_min_cost(c, qn);
            cap = cap > 2 * cost + 2 ? cap : 2 * cost + 2;
            cap = cap < best ? cap : best;
            cap = cap > cost ? cap : cost;
            u32 lb = 0;
            int r;
            const u32 slot = pool_find(c, pl, id);
            if (slot < c.pool) { /* where the step stopped was kept: on from there */
                AVK_LSTAT(20, 1);
                nodeA_replay_steps(c, n, code, depth);
                pool_load(c, pl, slot, n);
                r = nodeA_settle(c, n, cap, lb);
            } else {
                AVK_LSTAT(21, 1);
                if (nodeA_replay(c, n, code, depth - 1)) return LS_DEFER;
                r = nodeA_step(c, n, depth - 1, (code >> (2 * (depth - 1))) & 3u, cap, lb);
            }
            if (r == LS_DEFER) return LS_DEFER;
            if (r == LS_PARTIAL) {
                if (pushA(c, qn, lb, id, code, 1, depth)) return LS_DEFER;
                pool_keep(c, pl, id, n, true);
                continue;
            }
            const u32 full = nodeA_cost(n);
            if (full != cost) {
                if (pushA(c, qn, full, id, code, 0, depth)) return LS_DEFER;
                pool_keep(c, pl, id, n, false);
                continue;
            }
            have = true;
        }
        const u32 cnt = (u32)(bucket >> (8 * depth)) & 0xFFu;
        if (cnt >= c.max_branch) { /* :222 */
            if (!have && cost) {
                const u32 slot = pool_find(c, pl, id);
                if (slot < c.pool) pool_free(pl, slot);
            }
            continue;
        }
        bucket += 1ull << (8 * depth);
        if (!have) {
            if (cost == 0) nodeA_replay_zero(c, n, code, depth);
            else {
                /* the path says what was skipped; a cost that is all skipped calls leaves no distance: the fronts are the ends of the shorter sequences */
                nodeA_replay_steps(c, n, code, depth);
                const u32 slot = n.h[0].t_skip + n.h[0].q_skip + n.h[1].t_skip + n.h[1].q_skip == cost ? c.pool + 1u : pool_find(c, pl, id);
                if (slot > c.pool) {
                    n.h[0].d0 = n.h[0].t_len < n.h[0].q_len ? n.h[0].t_len : n.h[0].q_len;
                    n.h[1].d0 = n.h[1].t_len < n.h[1].q_len ? n.h[1].t_len : n.h[1].q_len;
                } else if (slot < c.pool) {
                    AVK_LSTAT(20, 1);
                    pool_load(c, pl, slot, n);
                } else {
                    AVK_LSTAT(21, 1);
                    if (nodeA_replay(c, n, code, depth)) return LS_DEFER;
                }
            }
        }
        if (depth == c.N) { /* :227-247 */
            const int r = nodeA_finalize(c, n, best);
            if (r == LS_DEFER) return LS_DEFER;
            if (r == LS_PARTIAL) continue; /* costs more than the best: neither kept nor tied */
            const u32 fc = nodeA_cost(n);
            if (fc < best) {
                best = fc;
                nbest = 0;
            }
            if (fc == best) {
                if (nbest >= c.optcap) return AVK_LDEFER(2);
                opt_set(c, nbest, code);
                nbest += 1;
            }
            continue;
        }
        const u32 slot = ord_slot(c, depth);
        const bool is_truth = slot < MV;
        const u32 zyg = (sel4(c.vw0, slot) >> 28) & 7u;
        const bool het = zyg == AVK_ZYG_UNPHASED_HET || zyg == AVK_ZYG_PHASED_HET01 || zyg == AVK_ZYG_PHASED_HET10;
        u32 lb = 0;
        if (het && (!is_truth || zyg == AVK_ZYG_UNPHASED_HET)) { /* :269-293: two clones, (REF|ALT) then (ALT|REF) */
            NodeA m = n;
            /* the first child works on the arrays of `n`; for the second the parent's fronts come back from a slot that held them meanwhile (or, without
             * a free slot, by a replay from the root) */
            const bool fronts = (m.h[0].ed | m.h[1].ed) != 0;
            u32 held = c.pool;
            if (fronts && c.pool) {
                for (u32 k = 0; k < c.pool; ++k)
                    if (((u32)(pl.ids >> (8 * k)) & 0xFFu) == 0xFFu) held = k;
                if (held < c.pool) pool_save(c, pl, held, 0xFEu, m); /* (an id no node has) */
            }
            int r = nodeA_step(c, n, depth, 2u, cost, lb);
            if (r == LS_DEFER) return LS_DEFER;
            if (pushA(c, qn, r ? lb : nodeA_cost(n), next_id, code | (2u << (2 * depth)), r ? 1u : 0u, depth + 1)) return LS_DEFER;
            pool_keep(c, pl, next_id, n, r != 0);
            next_id += 1;
            if (fronts) { /* parent wavefronts were overwritten */
                if (held < c.pool) pool_load(c, pl, held, m);
                else if (nodeA_replay(c, m, code, depth)) return LS_DEFER;
            }
            r = nodeA_step(c, m, depth, 1u, cost, lb);
            if (r == LS_DEFER) return LS_DEFER;
            if (pushA(c, qn, r ? lb : nodeA_cost(m), next_id, code | (1u << (2 * depth)), r ? 1u : 0u, depth + 1)) return LS_DEFER;
            pool_keep(c, pl, next_id, m, r != 0);
            next_id += 1;
        } else { /* :294-327: the node is moved, its id kept */
            u32 choice = 3u;
            if (het) choice = zyg == AVK_ZYG_PHASED_HET01 ? 2u : 1u; /* 0|1: REF on haplotype 1, ALT on haplotype 2 */
            const int r = nodeA_step(c, n, depth, choice, cost, lb);
            if (r == LS_DEFER) return LS_DEFER;
            if (pushA(c, qn, r ? lb : nodeA_cost(n), id, code | (choice << (2 * depth)), r ? 1u : 0u, depth + 1)) return LS_DEFER;
            pool_keep(c, pl, id, n, r != 0);
        }
        if (next_id > c.max_nodes) return AVK_LDEFER(3);
    }
    if (nbest == 0) return -100 - AVK_ST_NO_RESULTS; /* :331 */
    best_out = best;
    return (int)nbest;
}

/* ---- phase B: optimize_gt_alleles for one haplotype ------------------------------------------------------------------ */
/* The aligner of an ExactMatchNode has max_edit_distance 0 (:380): a live node is `exact`, its wavefront the single offset d0.
 * code: bit d = the allele assigned at depth d is ALT.  Entry: errors << 28 | (15 - (depth - errors)) << 24 | id << 12 | code << 4 | depth
 * (Reverse(errors), set - errors, Reverse(id); :452-458). */
AVK_DEV bool hapB_step(const LCtx &c, Hap &h, u32 d, bool alt) {
    const u32 slot = ord_slot(c, d);
    const bool ok = hap_step(c, h, slot < MV, true, slot, alt ? L_ALT : L_REF, sync_after(c, d));
    const u32 st = seq_id(c, 0, h.t_alt), sq = seq_id(c, 1, h.q_alt);
    h.d0 += match_run_same(c, st, h.t_len, sq, h.q_len, h.d0);
    const u32 lim = h.t_len < h.q_len ? h.t_len : h.q_len;
    return ok && h.d0 >= lim;
}
AVK_DEV u32 keyB(u32 errors, u32 depth, u32 id, u32 code) { return (errors << 28) | ((15u - (depth - errors)) << 24) | (id << 12) | (code << 4) | depth; }
/* in_t / in_q: the haplotype's input alleles (bit j = ALT for the side's call j).  Returns the number of flips and the final
 * alleles, LS_DEFER, or -100 - status. */
AVK_DEV int phaseB(const LCtx &c, u32 in_t, u32 in_q, u32 &res_t, u32 &res_q) {
    u32 qn = 0;
    q_push(c, qn, keyB(0, 0, 0, 0));
    u32 next_id = 1, min_sync = 0;
    while (qn > 0) {
        const u32 e = q_pop_min(c, qn);
        const u32 errors = e >> 28, depth = e & 0xFu, code = (e >> 4) & 0xFFu, id = (e >> 12) & 0xFFFu;
        Hap h;
        hap_init(h);
        AVK_LSTAT(14, 1);
        AVK_LSTAT(15, depth);
        /* every queued node is exact (that is what let it in): its wavefront is the end of its shorter sequence, no comparing again */
        for (u32 d = 0; d < depth; ++d) {
            const u32 slot = ord_slot(c, d);
            hap_step(c, h, slot < MV, true, slot, ((code >> d) & 1u) ? L_ALT : L_REF, sync_after(c, d));
        }
        h.d0 = h.t_len < h.q_len ? h.t_len : h.q_len;
        if (depth == c.N) { /* :180-192 */
            hap_step(c, h, true, false, 0, L_REF, c.L);
            const u32 st = seq_id(c, 0, h.t_alt), sq = seq_id(c, 1, h.q_alt);
            h.d0 += match_run_same(c, st, h.t_len, sq, h.q_len, h.d0);
            if (h.d0 >= h.t_len && h.d0 >= h.q_len) {
                res_t = h.t_alt;
                res_q = h.q_alt;
                return (int)errors;
            }
            continue;
        }
        if (depth < min_sync) continue;                                   /* :194-197 */
        if (h.t_len == h.q_len && h.t_refpos == h.q_refpos) min_sync = depth; /* :206-217 (fewer than 500 expansions: no auto-fail) */
        const u32 slot = ord_slot(c, depth);
        const bool is_truth = slot < MV;
        const bool cur_alt = (((is_truth ? in_t : in_q) >> (is_truth ? slot : slot - MV)) & 1u) != 0;
        if (!cur_alt) { /* :257-273 */
            if (hapB_step(c, h, depth, false)) {
                if (q_push(c, qn, keyB(errors, depth + 1, id, code))) return LS_DEFER;
            }
        } else { /* :274-306: (REF, error) first, then ALT */
            Hap r = h;
            if (hapB_step(c, r, depth, false)) {
                if (q_push(c, qn, keyB(errors + 1, depth + 1, next_id, code))) return LS_DEFER;
            }
            next_id += 1;
            if (hapB_step(c, h, depth, true)) {
                if (q_push(c, qn, keyB(errors, depth + 1, next_id, code | (1u << depth)))) return LS_DEFER;
            }
            next_id += 1;
        }
        if (next_id > 4000) return AVK_LDEFER(3);
    }
    return -100 - AVK_ST_NO_GT_RESULT; /* :345-348 */
}

/* What ONE call does to a string that has the window's bases over the call's REF span, decided from the window itself — never from the
 * caller's REF allele or its alt_ed, which may disagree with the genome (a record whose REF is not what the reference has there: the solver
 * splices the ALT into the WINDOW, generate_allele_sequence waffle_solver.rs:726-778, so that is what counts):
 *   one base for one base: 0 when the ALT base is the window's base, else 1 (equal lengths, one position);
 *   one side of the call is a single base and it is the window's base there (the anchor is kept): the call only inserts or only deletes the
 *   rest, the lengths differ by that much and that many edits suffice;
 *   anything else: -1, to be aligned. */
template <class C> AVK_DEV int call_effect(const C &c, u32 side, u32 j) {
    const u32 slot = (u32)C::MV * side + j;
    const u32 w0 = c.vw0_at(slot);
    const u32 pos = w0 & 0xFFu, a0 = (w0 >> 8) & 0xFFu, a1 = (w0 >> 16) & 0xFFu;
    if (a1 == 0 || a0 == 0) return -1;
    /* FULL(side, this call alone) has the ALT allele at `pos` (a single call is never dropped) */
    const bool anchored = ((extract16(c, 0, pos) ^ extract16(c, c.seq_id(side, 1u << j), pos)) & 3u) == 0;
    if (a0 == 1 && a1 == 1) return anchored ? 0 : 1;
    if ((a0 == 1 || a1 == 1) && anchored) return (int)(a0 > a1 ? a0 - a1 : a1 - a0);
    return -1;
}

/* wfa_ed(reference window, FULL(side, mask)).  No alleles: 0.  One call: call_effect.  Several calls, all applied: substitutions at two
 * different positions add up (equal lengths, the positions where the strings differ; no single edit gives two), pure insertions only (or pure
 * deletions only) add up as well (the length changes by their sum, which is also enough).  Anything else is aligned. */
template <class C> AVK_DEV int ed_to_ref(const C &c, u32 side, u32 mask, u32 len) {
    if (mask == 0) return 0;
    if ((mask & (mask - 1)) == 0) {
        const int e = call_effect(c, side, (u32)__builtin_ctz(mask));
        if (e >= 0) return e;
    } else if (c.seq_fail(c.seq_id(side, mask)) == 0) {
        u32 n = 0, n_snv = 0, n_ins = 0, n_del = 0, sum = 0, pos_x = 0;
        bool all = true;
#pragma unroll
        for (u32 j = 0; j < (u32)C::MV; ++j) {
            if (!((mask >> j) & 1u)) continue;
            const u32 w0 = c.vw0_side(side, j);
            const u32 a0 = (w0 >> 8) & 0xFFu, a1 = (w0 >> 16) & 0xFFu;
            const int e = call_effect(c, side, j);
            all = all && e >= 0;
            n += 1;
            sum += e >= 0 ? (u32)e : 0u;
            n_snv += (a0 == 1 && a1 == 1) ? 1u : 0u;
            n_ins += (a0 == 1 && a1 > 1) ? 1u : 0u;
            n_del += (a1 == 1 && a0 > 1) ? 1u : 0u;
            pos_x ^= w0 & 0xFFu;
        }
        if (all && n == 2 && n_snv == 2 && pos_x != 0) return (int)sum;
        if (all && (n_ins == n || n_del == n)) return (int)sum;
    }
    return wfa_ed(c, 0, c.L, c.seq_id(side, mask), len);
}

/* Distance between a haplotype string and the same string without ONE of its calls (slot), when call_effect decides it.  -1: not decided this way. */
template <class C> AVK_DEV int one_call_distance(const C &c, u32 slot) { return call_effect(c, slot / (u32)C::MV, slot % (u32)C::MV); }

/* the genotype assignment of one haplotype of an optimum: flips, observed alleles */
AVK_DEV int gt_for_hap(const LCtx &c, const Hap &h, u32 &rt, u32 &rq) {
    if (h.ed == 0 && h.nskip == 0) { /* truth == query with every ALT incorporated: the zero-flip path wins (exact_gt_optimizer.rs:169-192) */
        rt = h.t_alt;
        rq = h.q_alt;
        return 0;
    }
    rt = rq = 0;
    return phaseB(c, h.t_alt, h.q_alt, rt, rq);
}

/* ---- region staging -------------------------------------------------------------------------------------------------- */
struct SeqWriter {
    u64 acc;
    u32 nb, row, k;
};
template <class C> AVK_DEV void sw_push(const C &c, SeqWriter &w, u32 bits, u32 nbases) {
    if (nbases == 0) return;
    const u64 m = nbases >= 16 ? 0xFFFFFFFFull : ((1ull << (2 * nbases)) - 1ull);
    w.acc |= ((u64)bits & m) << w.nb;
    w.nb += 2 * nbases;
    if (w.nb >= 32) {
        *c.seq_word(w.row, w.k) = (u32)w.acc;
        w.k += 1;
        w.acc >>= 32;
        w.nb -= 32;
    }
}
template <class C> AVK_DEV void sw_ref(const C &c, SeqWriter &w, u32 from, u32 to) {
    for (u32 p = from; p < to; p += 16) sw_push(c, w, extract16(c, 0, p), to - p < 16 ? to - p : 16u);
}
/* FULL(side, mask): the calls of the mask applied in the side's order, a call that starts before the end of the previous applied one
 * is dropped and its alt_ed counted (generate_allele_sequence, waffle_solver.rs:726-778) */
template <u32 side> AVK_DEV void build_full(LCtx &c, u32 mask, const u32 (&a1lo)[NS], const u32 (&a1hi)[NS]) {
    const u32 s = seq_id(c, side, mask), cnt = side == 0 ? c.T : c.Q;
    SeqWriter w;
    w.acc = 0;
    w.nb = 0;
    w.row = s;
    w.k = 0;
    u32 cur = 0, len = 0, failed = 0;
#pragma unroll
    for (u32 j = 0; j < MV; ++j) { /* static slot indices: the records stay in registers */
        if (j >= cnt || !((mask >> j) & 1u)) continue;
        const u32 slot = MV * side + j;
        const u32 w0 = c.vw0[slot];
        const u32 pos = w0 & 0xFFu, a0 = (w0 >> 8) & 0xFFu, a1 = (w0 >> 16) & 0xFFu;
        if (pos < cur) {
            failed += c.vw1[slot] & 0xFFu;
            continue;
        }
        sw_ref(c, w, cur, pos);
        sw_push(c, w, a1lo[slot], a1 < 16 ? a1 : 16u);
        if (a1 > 16) sw_push(c, w, a1hi[slot], a1 - 16);
        len += pos - cur + a1;
        cur = pos + a0;
    }
    if (cur < c.L) {
        sw_ref(c, w, cur, c.L);
        len += c.L - cur;
    }
    if (w.nb) *c.seq_word(w.row, w.k) = (u32)w.acc;
    const u64 len_s = (u64)len << (8 * (s & 7u)), fail_s = (u64)failed << (8 * (s & 7u));
    c.seq_len_lo |= s < 8 ? len_s : 0ull;
    c.seq_len_hi |= s < 8 ? 0ull : len_s;
    c.seq_fail_lo |= s < 8 ? fail_s : 0ull;
    c.seq_fail_hi |= s < 8 ? 0ull : fail_s;
}

/* The reference window of a region into row 0 of its table: 2 bits per base from the packed genome, word k by lane q of NQ when k = q (mod NQ).
 * Every global load is issued before the first one is used — a lane is otherwise a chain of memory round trips here: the flags of the packed words the
 * window touches (at most 14 consecutive bits of the bitmap: two words) and the words themselves (static loop: at most 13 words of 16 bases).
 * Returns false when a flagged word (anything but upper-case ACGT) is in the window: not for these kernels. */
template <u32 NQ> AVK_DEV bool load_window(const AvkKernelArgs &a, const LCtx &c, u32 h0, u32 shift, u32 q) {
    enum { KMAX = 13, NJ = (KMAX + NQ - 1) / NQ };
    const u32 nw = (c.L + shift + 15u) >> 4; /* packed words the window touches */
    const u64 w0 = h0;
    const u32 e0 = a.ref_exc[w0 >> 5], e1 = a.ref_exc[(w0 + nw - 1) >> 5];
    u32 lo[NJ + 1], hi[NJ];
#pragma unroll
    for (u32 j = 0; j < NJ; ++j) {
        const u32 k = q + NQ * j;
        const bool on = k < c.W1 && k * 16 < c.L;
        /* (one lane for all words: the word behind word k is the next load, wanted whenever word k is) */
        const bool need = NQ > 1 ? on : (k == 0 || (k - 1u < c.W1 && (k - 1u) * 16 < c.L));
        lo[j] = a.ref_2bit[w0 + (need ? k : 0u)];
        if (NQ > 1) hi[j] = a.ref_2bit[w0 + (on ? k + 1u : 0u)];
    }
    if (NQ == 1) lo[NJ] = a.ref_2bit[w0 + ((u32)KMAX - 1u < c.W1 && ((u32)KMAX - 1u) * 16 < c.L ? (u32)KMAX : 0u)];
    const u64 flags = ((((u64)e1 << 32) | e0) >> (w0 & 31u)) & ((1ull << nw) - 1ull);
    if (flags) return false;
#pragma unroll
    for (u32 j = 0; j < NJ; ++j) {
        const u32 k = q + NQ * j;
        if (k >= c.W1) continue;
        const u32 h = NQ > 1 ? hi[j] : lo[j + 1];
        c.p[k << c.ls] = k * 16 < c.L ? (u32)((((u64)h << 32) | lo[j]) >> (2 * shift)) : 0u;
    }
    return true;
}

/* ---- one region ------------------------------------------------------------------------------------------------------ */
struct LaneOut {
    u32 ed1, ed2, n_opt, present;
};
/* 22 counters of one metric group, in the field order of include/aardvark_amd.h */
struct Group22 {
    u32 f[AVK_N_FIELDS];
};
/* GroupMetrics::add_truth_zygosity (grouped_metrics.rs:183-227); query calls are scored as truth and land in the query columns
 * (add_swap_benchmark, grouped_metrics.rs:268-277) */
template <bool q> AVK_DEV void g_add(Group22 &g, u32 w, u32 exp, u32 obs) {
    const int f_gt_tp = q ? AVK_F_GT_QUERY_TP : AVK_F_GT_TRUTH_TP, f_gt_fn = q ? AVK_F_GT_QUERY_FP : AVK_F_GT_TRUTH_FN;
    const int f_gt_fn_gt = q ? AVK_F_GT_QUERY_FP_GT : AVK_F_GT_TRUTH_FN_GT;
    const int f_hap_tp = q ? AVK_F_HAP_QUERY_TP : AVK_F_HAP_TRUTH_TP, f_hap_fn = q ? AVK_F_HAP_QUERY_FP : AVK_F_HAP_TRUTH_FN;
    const int f_w_tp = q ? AVK_F_WHAP_QUERY_TP : AVK_F_WHAP_TRUTH_TP, f_w_fn = q ? AVK_F_WHAP_QUERY_FP : AVK_F_WHAP_TRUTH_FN;
    /* branch-free form of the two cases (equal: everything expected was observed; fewer: the difference is missed) */
    const u32 eq = exp == obs ? 1u : 0u;
    g.f[f_hap_tp] += obs;
    g.f[f_hap_fn] += exp - obs;
    g.f[f_w_tp] += obs * w;
    g.f[f_w_fn] += (exp - obs) * w;
    g.f[f_gt_tp] += eq;
    g.f[f_gt_fn] += 1u - eq;
    g.f[f_gt_fn_gt] += (1u - eq) & (obs > 0 ? 1u : 0u);
}

struct LaneArgs { /* what a launch of the lane kernel needs besides AvkKernelArgs' outputs */
    const u32 *recs;     /* the class's fast records, tile-major: word w of lane l of tile t at recs[(t * rec_words + w) * 64 + l] */
    u32 rec_words;       /* AVK_FAST_WORDS_OF(calls per side of the class) */
    u32 n_tiles;
    u32 *tile_counter;
    u32 W, nm, ed_max, qcap; /* the class of this launch */
    u32 gen_base;        /* record index (work order of the wave-per-region kernels) of fast record 0 */
    u32 max_ed_c;        /* largest distance a lane follows in the alignments of the metrics phase (beyond: handed over); 0 = what the rows hold */
    u32 max_nodes;       /* phase A gives up (hands the region over) beyond this many search nodes: at most 250 (ids are 8 bits) */
    u32 pool;            /* node states a lane keeps during its search (NodePool; at most 8, 0 = every pop replays its path) */
    u32 lanes_log2;      /* 6, 5 or 4: a wave takes 64, 32 or 16 records of a tile at a time on its first lanes (smaller LDS slice per wave, more
                            waves per CU, less waiting for the slowest record) */
};

/* returns AVK_ST_* (>= 0) or LS_DEFER.  `tally` = the workgroup's LDS tally (u32 counters). */
AVK_DEV int solve_lane(const AvkKernelArgs &a, LCtx &c, const u32 *rec, u32 lane_stride, LaneOut &out, u32 *tally) {
    const u32 h0 = rec[0], h1 = rec[1 * lane_stride], v_off = rec[2 * lane_stride], orig = rec[3 * lane_stride];
    const u32 shift = h1 & 15u;
    c.L = (h1 >> 4) & 0xFFu;
    c.T = (h1 >> 12) & 3u;
    c.Q = (h1 >> 14) & 3u;
    c.N = c.T + c.Q;
    { /* the record says which side each search depth takes its call from; the sides' calls come in their own order */
        const u32 takeq = h1 >> 16;
        u32 ord = 0, it = 0, iq = 0;
        for (u32 d = 0; d < c.N; ++d) {
            const u32 q = (takeq >> d) & 1u;
            ord |= (q ? MV + iq : it) << (3 * d);
            iq += q;
            it += 1u - q;
        }
        c.ord = ord;
        c.takeq = takeq & ((1u << c.N) - 1u);
    }
    c.max_branch = a.max_branch_factor;
    c.seq_len_lo = c.L;
    c.seq_len_hi = c.seq_fail_lo = c.seq_fail_hi = 0;
    u32 a1lo[NS], a1hi[NS];
    u32 types = 0;
    const u32 maxv = c.nm1 == 1 ? 1u : (c.nm1 == 3 ? 2u : 3u); /* record slots: truth [0, maxv), query [maxv, 2 maxv) */
#pragma unroll
    for (u32 s = 0; s < NS; ++s) {
        const u32 side = s / MV, j = s % MV;
        const bool on = j < (side ? c.Q : c.T);
        c.vw0[s] = c.vw1[s] = a1lo[s] = a1hi[s] = 0;
        if (j < maxv) { /* every slot the class's records have is loaded, whatever the header says: one round trip for the whole record */
            const u32 *v = rec + (AVK_FAST_HDR + 4 * (side * maxv + j)) * lane_stride;
            const u32 x0 = v[0], x1 = v[1 * lane_stride], x2 = v[2 * lane_stride], x3 = v[3 * lane_stride];
            c.vw0[s] = on ? x0 : 0u;
            c.vw1[s] = on ? x1 : 0u;
            a1lo[s] = on ? x2 : 0u;
            a1hi[s] = on ? x3 : 0u;
            types |= on ? 1u << ((x0 >> 24) & 0xFu) : 0u;
        }
    }
    { /* tables of the search: sync points per depth, ends of the REF spans per side, whether a cost of 0 means that nothing was skipped */
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
    /* reference window: 2 bits per base from the packed genome; a flagged word (anything but upper-case ACGT) is not for this kernel */
    if (!load_window<1>(a, c, h0, shift, 0)) return AVK_LDEFER(4);
    for (u32 m = 1; m <= c.nm1; ++m) {
        if (m < (1u << c.T)) build_full<0>(c, m, a1lo, a1hi);
        if (m < (1u << c.Q)) build_full<1>(c, m, a1lo, a1hi);
    }
    AVK_LT_MARK(c, 0)

    /* ---- phase A */
    AVK_LSTAT(5, 1);
    AVK_LPHASE(0);
    u32 best_cost = 0;
    const int nopt = phaseA(c, best_cost);
    AVK_LT_MARK(c, 1)
    if (nopt == LS_DEFER) return LS_DEFER;
    if (nopt < 0) return -nopt - 100;
    out.n_opt = (u32)nopt;
    if (a.mode == 1) { /* merge_solver.rs:137-143 */
        out.ed1 = best_cost == 0 ? 1u : 0u;
        return AVK_ST_OK;
    }

    /* ---- phase B for every tied optimum (waffle_solver.rs:169-261); the first optimum with the fewest flips wins (:264-265) */
    AVK_LPHASE(1);
    u32 best_total = 0xFFFFFFFFu;
    u32 o_t0 = 0, o_q0 = 0, o_t1 = 0, o_q1 = 0; /* observed alleles of the winner, per haplotype and side */
    NodeA wn;
    for (u32 k = 0; k < (u32)nopt; ++k) {
        const u32 code = opt_get(c, k);
        if (k) { /* the mirror image of an earlier optimum — the same two haplotypes the other way round, as every orientation of unphased hets has one — needs the
                    same flips in total and comes later: it cannot win (:264-265 takes the first of the fewest), and it fails where the earlier one would have */
            const u32 mirror = ((code & 0x555u) << 1) | ((code >> 1) & 0x555u);
            bool seen = false;
            for (u32 j = 0; j < k && mirror != code; ++j) seen = seen || opt_get(c, j) == mirror;
            if (seen) continue;
        }
        NodeA n;
        if (best_cost == 0) { /* nothing skipped, no edits: the finished haplotypes are equal sequences */
            nodeA_replay_zero(c, n, code, c.N);
            hap_step(c, n.h[0], true, false, 0, L_REF, c.L);
            hap_step(c, n.h[1], true, false, 0, L_REF, c.L);
            n.h[0].d0 = n.h[0].t_len;
            n.h[1].d0 = n.h[1].t_len;
        } else {
            if (nodeA_replay(c, n, code, c.N)) return LS_DEFER;
            if (nodeA_finalize(c, n)) return LS_DEFER;
        }
        u32 rt0, rq0, rt1, rq1;
        const int e0 = gt_for_hap(c, n.h[0], rt0, rq0);
        if (e0 == LS_DEFER) return LS_DEFER;
        if (e0 < 0) return -e0 - 100;
        const int e1 = gt_for_hap(c, n.h[1], rt1, rq1);
        if (e1 == LS_DEFER) return LS_DEFER;
        if (e1 < 0) return -e1 - 100;
        const u32 total = (u32)e0 + (u32)e1;
        if (total < best_total) {
            best_total = total;
            wn = n;
            o_t0 = rt0;
            o_q0 = rq0;
            o_t1 = rt1;
            o_q1 = rq1;
            if (total == 0) break;
        }
    }
    out.ed1 = wn.h[0].ed;
    out.ed2 = wn.h[1].ed;
    AVK_LT_MARK(c, 2)

    /* ---- phase C: compare_expected_observed (:296-327) + per-call outputs */
    u32 exp_pack = 0, obs_pack = 0; /* 2 bits per call slot */
    int bad = 0;
#pragma unroll
    for (u32 s = 0; s < NS; ++s) {
        const u32 side = s / MV, j = s % MV;
        const bool on = j < (side ? c.Q : c.T);
        const u32 b0 = ((side ? wn.h[0].q_alt : wn.h[0].t_alt) >> j) & 1u, b1 = ((side ? wn.h[1].q_alt : wn.h[1].t_alt) >> j) & 1u;
        const u32 o0 = ((side ? o_q0 : o_t0) >> j) & 1u, o1 = ((side ? o_q1 : o_t1) >> j) & 1u;
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
        a.var_out[v_off + (side ? c.T + j : j)] = ea | (oa << 8) | (cls << 16) | (rz << 24);
    }
    if (bad) return bad;

    /* add_basepair_stats (:335-449): per haplotype X = 2 ed(ref, truth), Y = 2 ed(ref, query), Z = 2 ed(truth, query) */
    const u32 SUPMASK = (1u << AVK_VT_SNV) | (1u << AVK_VT_INSERTION) | (1u << AVK_VT_DELETION) | (1u << AVK_VT_INDEL) | (1u << AVK_VT_TR_CONTRACTION) |
                        (1u << AVK_VT_TR_EXPANSION) | (1u << AVK_VT_SV_DELETION) | (1u << AVK_VT_SV_INSERTION);
    out.present = types | SUPMASK;
    u32 X0 = 0, Y0 = 0, tp0 = 0, X1 = 0, Y1 = 0, tp1 = 0;
    AVK_LPHASE(2);
    /* a second haplotype with the alleles of the first is the same pair of sequences: same distances */
    const bool same_haps = wn.h[0].t_alt == wn.h[1].t_alt && wn.h[0].q_alt == wn.h[1].q_alt;
    for (u32 hh = 0; hh < 2; ++hh) {
        const Hap &h = hh ? wn.h[1] : wn.h[0];
        if (hh && same_haps) {
            X1 = X0;
            Y1 = Y0;
            tp1 = tp0;
            break;
        }
        int ert = 0, erq = 0;
        if (h.t_alt) {
            ert = ed_to_ref(c, 0, h.t_alt, h.t_len);
            if (ert < 0) return LS_DEFER;
        }
        if (h.q_alt) {
            if (h.ed == 0 && h.t_alt) erq = ert;
            else {
                erq = ed_to_ref(c, 1, h.q_alt, h.q_len);
                if (erq < 0) return LS_DEFER;
            }
        } else if (h.ed == 0) erq = ert;
        const u32 Xh = 2u * (u32)ert, Yh = 2u * (u32)erq, tph = (Xh + Yh - 2u * h.ed) / 2u;
        if (hh) {
            X1 = Xh;
            Y1 = Yh;
            tp1 = tph;
        } else {
            X0 = Xh;
            Y0 = Yh;
            tp0 = tph;
        }
    }
    AVK_LT_MARK(c, 3)
    /* Alignments of the per-type groups (:383-445), done BEFORE anything is added to the tally (they can still hand the region over).
     * One pass over the call types of the region: a side that has calls of the type AND calls of other types is compared with only the
     * type's calls applied, per haplotype: x = ed(ref, filtered side), z = ed(filtered side, other side as it is), 8 bits each.  The
     * pair goes to the rows of the optima list (read and done with), entry (hap, side, first call of the type on the side), for the
     * pass below that adds the groups up. */
    AVK_LPHASE(3);
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
        for (u32 side = 0; side < 2; ++side) {
            const u32 mask_g = side ? qmask_g : tmask_g;
            if (mask_g == 0 || mask_g == (side ? q_all : t_all)) continue; /* none of the type, or nothing but the type: no filtering */
            for (u32 hh = 0; hh < 2; ++hh) {
                const Hap &h = hh ? wn.h[1] : wn.h[0];
                const u32 e_idx = (hh * 2 + side) * MV + (u32)__builtin_ctz(mask_g);
                if (hh && same_haps) {
                    filt_set(c, e_idx, filt_get(c, e_idx - 2 * MV));
                    continue;
                }
                const u32 st = seq_id(c, 0, h.t_alt), sq = seq_id(c, 1, h.q_alt);
                const u32 Xh = hh ? X1 : X0, Yh = hh ? Y1 : Y0;
                const u32 alt = side ? h.q_alt : h.t_alt, m = alt & mask_g;
                u32 x2 = 0, z2 = (side ? Xh : Yh) / 2; /* nothing left of the side: it is the reference window */
                if (m && m == alt) { /* nothing filtered away on this haplotype: the side as it is */
                    x2 = (side ? Yh : Xh) / 2;
                    z2 = h.ed;
                } else if (m) {
                    const u32 sf = seq_id(c, side, m), fl = seq_len_of(c, sf);
                    const int x = ed_to_ref(c, side, m, fl);
                    /* equal haplotypes with everything applied, one call filtered away: the other side's string is the filtered string plus that call */
                    const u32 gone = alt ^ m;
                    int z = -1;
                    if (h.ed == 0 && h.nskip == 0 && (gone & (gone - 1)) == 0 && seq_fail_of(c, side ? sq : st) == 0 && seq_fail_of(c, sf) == 0)
                        z = one_call_distance(c, MV * side + (u32)__builtin_ctz(gone));
                    if (z < 0) z = side ? wfa_ed(c, st, h.t_len, sf, fl) : wfa_ed(c, sf, fl, sq, h.q_len);
                    if (x < 0 || z < 0) return LS_DEFER;
                    x2 = (u32)x;
                    z2 = (u32)z;
                }
                filt_set(c, e_idx, x2 | (z2 << 8));
            }
        }
    }
    AVK_LT_MARK(c, 4)
    /* groups that can hold anything: the joint one and one per call type of the region */
    u32 *gm_out = a.group_metrics ? a.group_metrics + (u64)orig * AVK_N_GROUPS * AVK_N_FIELDS : (u32 *)0;
    if (gm_out)
        for (u32 i = 0; i < AVK_N_GROUPS * AVK_N_FIELDS; ++i) gm_out[i] = 0;
    u32 *bp_dst = a.bp_out ? a.bp_out + 4 * (u64)a.bp_off[orig] : (u32 *)0; /* compact BASEPAIR groups: the loop below visits them in their order */
    /* The joint group comes first (bit 0): its RECORD_BP check is the only way the region can still fail, so nothing has been added to
     * the tally when it does; every alignment has been made above. */
    for (u32 left = 1u | (types << 1); left; left &= left - 1) {
        const u32 g = (u32)__builtin_ctz(left);
        Group22 G;
#pragma unroll
        for (int i = 0; i < AVK_N_FIELDS; ++i) G.f[i] = 0;
        u32 tot_t = 0, tot_q = 0, tcount = 0, qcount = 0;
        u32 tmask_g = 0, qmask_g = 0; /* the side's calls of this group's type */
#pragma unroll
        for (u32 s = 0; s < NS; ++s) {
            const bool on = (s % MV) < (s < MV ? c.T : c.Q);
            const u32 w0 = c.vw0[s], w1 = c.vw1[s];
            const u32 vt = (w0 >> 24) & 0xFu, z = (w0 >> 28) & 7u;
            if (!on || (g != 0 && vt != g - 1)) continue;
            if (s >= MV) g_add<true>(G, w1 & 0xFFu, (exp_pack >> (2 * s)) & 3u, (obs_pack >> (2 * s)) & 3u); /* static field indices */
            else g_add<false>(G, w1 & 0xFFu, (exp_pack >> (2 * s)) & 3u, (obs_pack >> (2 * s)) & 3u);
            const u32 cntz = z == AVK_ZYG_HOM_ALT ? 2u : ((z == AVK_ZYG_UNPHASED_HET || z == AVK_ZYG_PHASED_HET01 || z == AVK_ZYG_PHASED_HET10) ? 1u : 0u);
            const u32 val = cntz * ((w1 >> 8) & 0xFFFFu);
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
            G.f[AVK_F_BP_TRUTH_FN] += X0 - tp0 + 2 * wn.h[0].t_skip + X1 - tp1 + 2 * wn.h[1].t_skip; /* + skip metrics :378-381 */
            G.f[AVK_F_BP_QUERY_TP] += tp0 + tp1;
            G.f[AVK_F_BP_QUERY_FP] += Y0 - tp0 + 2 * wn.h[0].q_skip + Y1 - tp1 + 2 * wn.h[1].q_skip;
        } else if ((SUPMASK >> (g - 1)) & 1u) { /* :383-445: one side filtered to the type against the other side as it is */
            for (u32 hh = 0; hh < 2; ++hh) {
                const Hap &h = hh ? wn.h[1] : wn.h[0];
                const u32 Xh = hh ? X1 : X0, Yh = hh ? Y1 : Y0, tph = hh ? tp1 : tp0;
                u32 q_tp = 0, q_fp = 0, t_tp = 0, t_fn = 0;
                if (qcount) {
                    if (qcount == c.Q) {
                        q_tp = tph;
                        q_fp = Yh - tph + 2 * h.q_skip;
                    } else { /* some of the side's calls: the alignments were made above */
                        const u32 sf = seq_id(c, 1, h.q_alt & qmask_g);
                        const u32 e = filt_get(c, (hh * 2 + 1u) * MV + (u32)__builtin_ctz(qmask_g));
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
                        const u32 e = filt_get(c, (hh * 2) * MV + (u32)__builtin_ctz(tmask_g));
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
#pragma unroll
        for (int i = 0; i < AVK_N_FIELDS; ++i) {
            const u32 v = G.f[i];
            if (!v) continue;
            avk_tally_add_u32(tally, g * AVK_N_FIELDS + i, v); /* lanes with the same (counter, value) share one LDS atomic */
            if (gm_out) gm_out[g * AVK_N_FIELDS + i] = v;
        }
        if (bp_dst) {
            avk_u4 w;
            w.x = G.f[AVK_F_BP_TRUTH_TP], w.y = G.f[AVK_F_BP_TRUTH_FN], w.z = G.f[AVK_F_BP_QUERY_TP], w.w = G.f[AVK_F_BP_QUERY_FP];
            *(avk_u4 *)bp_dst = w;
            bp_dst += 4;
        }
    }
    return AVK_ST_OK;
}

/* LDS words per wave for a launch class */
AVK_DEV u32 lane_optcap(u32 nm) { return nm == 2 ? 8u : 16u; } /* tied optima a lane keeps (16 bits each); more: handed over */
AVK_DEV u32 lane_pool_default(u32 nm) { return nm == 2 ? 2u : (nm == 4 ? 4u : 6u); } /* kept node states per lane (NodePool), by calls per side */
AVK_DEV u32 lane_rows(u32 W, u32 nm, u32 ed_max, u32 qcap, u32 pool) {
    const u32 ns = 1 + 2 * (nm - 1);
    const u32 wfr = (2 * ed_max + 2 + 3) / 4;
    return ns * (W + 1) + 3 * wfr + qcap + lane_optcap(nm) / 2 + pool * 2 * wfr;
}

/* One persistent wave: claims tiles of 64 fast records, every lane solves its record.  wave_lds = this wave's rows, wg_tally =
 * the workgroup's 288 tally words in LDS (zeroed by the caller, flushed by the caller). */
AVK_DEV void lane_worker(const AvkKernelArgs &a, const LaneArgs &la, u32 wave_id, u32 *wave_lds, u32 *wg_tally, u32 &n_ok_out, u32 &n_err_out, u64 *part = (u64 *)0) {
    const u32 lane = (u32)wv_lane();
    LCtx c;
    c.ls = la.lanes_log2;
    c.p = wave_lds + (lane & ((1u << la.lanes_log2) - 1u));
    c.W1 = la.W + 1;
    c.nm1 = la.nm - 1;
    c.off_wf = (1 + 2 * c.nm1) * c.W1;
    c.wfr = (2 * la.ed_max + 2 + 3) / 4;
    c.wfcap = 2 * la.ed_max + 2;
    c.off_q = c.off_wf + 3 * c.wfr;
    c.wfcap_c = 4 * (3 * c.wfr + la.qcap);
    if (la.max_ed_c && 2 * la.max_ed_c + 3 < c.wfcap_c) c.wfcap_c = 2 * la.max_ed_c + 3;
    c.qcap = la.qcap;
    c.off_opt = c.off_q + la.qcap;
    c.optcap = lane_optcap(la.nm);
    c.off_pool = c.off_opt + c.optcap / 2;
    c.pool = la.pool < 8u ? la.pool : 8u;
    c.max_nodes = la.max_nodes < 250u ? la.max_nodes : 250u;
    u32 n_ok = 0, n_err = 0, n_tiles_done = 0;
#ifdef AVK_LANE_PHASE_TIMING
    for (int k = 0; k < 8; ++k) c.tph[k] = 0;
#endif
    const u32 width = 1u << la.lanes_log2, parts = 64u >> la.lanes_log2; /* records per claim, claims per tile */
    const u32 n_claims = la.n_tiles * parts;
    for (;;) {
        u32 t = 0;
#ifdef AVK_LANE_PHASE_TIMING
        const u64 t_tile0 = avk_clock();
#endif
        if (lane == 0) t = avk_atomic_add_u32_global(la.tile_counter, 1u);
        t = wv_uni(wv_shfl(t, 0));
        if (t >= n_claims) break;
        n_tiles_done += 1;
        if (part && (n_tiles_done & 15u) == 0) { /* the 32-bit LDS tally moves on to the 64-bit partial tally every 16 claims (at most 1024 regions) */
            wv_sync();
            for (u32 i = lane; i < AVK_N_GROUPS * AVK_N_FIELDS; i += 64) {
                const u32 v = avk_wg_xchg(wg_tally + i, 0u);
                if (v) avk_atomic_add_u64_global(part + i, v);
            }
            wv_sync();
        }
        const u32 rl = (t & (parts - 1u)) * width + lane; /* this lane's record in the tile */
        t >>= 6u - la.lanes_log2;
        if (lane >= width) continue; /* the other lanes only take part in the wave's claims and flushes */
#ifdef AVK_LANE_SLOW_TILES
        c.n_pops = c.n_diag = c.n_words = 0;
#endif
        const u32 *rec = la.recs + (u64)t * la.rec_words * 64u + rl;
        const u32 h1 = rec[64];
        if (h1 != 0xFFFFFFFFu) { /* a lane of the class's last tile may have no region */
            const u32 orig = rec[3 * 64];
            const u32 v_off = rec[2 * 64];
            LaneOut out;
            out.ed1 = out.ed2 = out.n_opt = out.present = 0;
#ifdef AVK_LANE_PHASE_TIMING
            c.tlast = avk_clock();
#endif
#ifdef AVK_LANE_STATS
            const uint64_t work0 = lane_work_now();
            uint64_t comp0[16];
            for (int k = 0; k < 16; ++k) comp0[k] = g_lane_stats[k];
#endif
            const int st = solve_lane(a, c, rec, 64u, out, wg_tally);
#ifdef AVK_LANE_STATS
            if (g_lane_work) g_lane_work[orig] = (uint32_t)(lane_work_now() - work0);
            if (g_lane_comp)
                for (int k = 0; k < 16; ++k) g_lane_comp[16 * (size_t)orig + k] = (uint32_t)(g_lane_stats[k] - comp0[k]);
#endif
            if (st == AVK_ST_OK) {
                AVK_LT_MARK(c, 5)
            }
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
                    const u32 h1b = h1;
                    const u32 nvar = ((h1b >> 12) & 3u) + ((h1b >> 14) & 3u);
                    for (u32 k = 0; k < nvar; ++k) a.var_out[v_off + k] = 0;
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
        {
            const u64 dt_tile = avk_clock() - t_tile0;
            c.tph[6] += dt_tile;
#ifdef AVK_LANE_SLOW_TILES
            if (la.lanes_log2 < 6) { /* profiling: how long every narrow tile took (its first lane says), and which lanes did the most turns */
                if (lane == 0) printf("tile nm %u claim %u ticks %llu\n", la.nm, t * parts + (rl >> la.lanes_log2), (unsigned long long)dt_tile);
                if (c.n_diag > (u32)AVK_LANE_SLOW_TILES || c.n_pops > 60)
                    printf("busylane nm %u claim %u lane %u orig %u pops %u diagonals %u words %u\n", la.nm, t * parts + (rl >> la.lanes_log2), lane,
                           la.recs[((u64)t * la.rec_words + 3) * 64u + rl], c.n_pops, c.n_diag, c.n_words);
            }
#endif
        }
#endif
    }
#ifdef AVK_LANE_PHASE_TIMING
    if (lane < width && (la.nm > 4 || (la.nm == 4 && la.lanes_log2 < 6))) { /* the launches that end a step: the three-call class, the head of the two-call class */
        u64 *pc = a.tally + (u64)(wave_id % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE + AVK_TALLY_LEN + 5 + (la.nm > 4 ? 0 : 8);
        for (int k = 0; k < 7; ++k) avk_atomic_add_u64_global(pc + k, c.tph[k]);
        avk_atomic_add_u64_global(pc + 7, 1);
    }
#endif
    n_ok_out = n_ok;
    n_err_out = n_err;
}

} // namespace lane
} // namespace avk
#endif
