/*
 * avk_wide.inl — the compare solver for regions with LARGE SEARCHES on small windows: one region per wavefront, every lane a PIECE of the search.
 *
 * The regions that end a whole-genome step are not large: a window of ~150 bases with four or five calls on a side, most of them unphased
 * heterozygous, so that optimize_sequences (src/query_optimizer.rs:166-365) keeps dozens of orientations alive and pops 30-100 nodes.  The
 * wave-per-region kernel (avk_solver.inl) walks such a search one node at a time — copy the node, extend one haplotype, extend the other, each a chain
 * of dependent memory round trips on one wave: 13 us per node extension, 2.5 ms for the slowest regions.  The lane-per-region kernel (avk_lane.inl)
 * has the right arithmetic for them (2-bit sequences, a haplotype is a PREFIX of one of 2^calls full-length sequences per side, so nothing is ever
 * copied) but gives a region one lane.  This kernel combines the two:
 *
 *   - one region per wave, its tables in the wave's LDS: the call records, the 2^T + 2^Q - 1 full-length sequences FULL(side, mask) (built by the
 *     64 lanes side by side, generate_allele_sequence's rule, src/waffle_solver.rs:726-778), the search nodes (two packed haplotype states of 13
 *     words: positions, lengths, skip penalties, alleles, wavefront offsets as bytes);
 *   - the best-first queue is kept SORTED in registers, entry j in lane j: the next pops are lanes 0, 1, 2, ..;
 *   - a ROUND expands the first up to 16 queue entries that have not been expanded yet, all at once: lane (entry, child, haplotype) runs one
 *     HaplotypeDWFA::extend_variant (src/dwfa/haplotype_dwfa.rs:46-67) with avk_lane.inl's primitives — extending a node is a pure function of the
 *     node and its depth, so doing it before the node's turn cannot change its result;
 *   - the COMMIT loop then replays the reference's loop (:203-328) on the sorted queue with scalar bookkeeping only: cost > best ends the search,
 *     the per-depth quota drops or counts the node, a finished node updates the optima, an expanded node's children get their ids in the
 *     reference's order and are inserted; it stops at the first entry that has no expansion yet (a child that sorts in front of expanded entries
 *     simply postpones them: their expansions stay valid).  Pop order, ids, quota counts and the order of tied optima are the reference's.
 *   - optimize_gt_alleles (src/exact_gt_optimizer.rs:108-357) runs for all (tied optimum, haplotype) pairs at once, one search per lane; the
 *     alignments of the metrics phase (src/waffle_solver.rs:335-449) are one per lane, the metric groups one per lane.
 *
 * What does not fit (window + growth + edit bound over 255, more than 8 calls on a side, ALT alleles over 32 bases or with other symbols than
 * ACGT, a flagged reference word, a wavefront past W_ED_MAX on a node that must be resolved, queue / pool / optima capacities) is appended to the
 * launch's overflow list for the wave-per-region kernels: results never depend on which kernel solved a region.
 */
#ifndef AVK_WIDE_INL
#define AVK_WIDE_INL

#include "avk_dev_types.h"
#include "avk_wave.h"
#include "avk_lane.inl"
#include "avk_solver.inl" /* write_region_record, write_failed_region, dp_region_record */

namespace avk {
namespace wide {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
using lane::Hap;
using lane::L_ALT;
using lane::L_REF;

enum { WMV = 8, WNS = 16 };  /* call slots per side, per region: slot = WMV * side + index on the side */
enum { W_ED_MAX = AVK_WIDE_ED_MAX };      /* largest distance a search node's wavefront holds (2 ed + 1 offsets, one byte each); a region's wavefront blocks are sized by its own
                                edit bound (AvkDevRegion::ed_bound: no two strings of the region are further apart), which is what the wave-per-region kernel trusts too */
enum { W_NBLK_MAX = 254, W_NBLK_MIN = 8 }; /* wavefront blocks of a region (a state names its block in 8 bits) */
enum { W_NODE_W = 9 };       /* words of a search node: two haplotype states of 4 words (odd stride: spreads the banks) */
enum { W_K = 16 };           /* queue entries expanded per round (x up to 2 children x 2 haplotypes = the 64 lanes) */
enum { W_QR = 4 };           /* rows of the sorted queue: 64 entries each, entry 64 r + j in lane j of row r */
enum { W_POOL_MAX = 240 };   /* most search nodes alive (a node's slot is 8 bits) */
enum { W_POOL_MIN = 24 };
enum { W_OPTCAP = 64 };      /* tied optima kept */
enum { W_QB = 16 };          /* queue entries of one genotype search (4 words each) */
enum { W_QB_W = 4 * W_QB + 1 }; /* words of one genotype search's queue (odd: spreads the banks) */
enum { W_COOP_ED = 8 };      /* a search node's aligner is taken over by the whole wave when its distance passes this (dw_continue_wave) */
enum { W_COOP_MIN = 12 };    /* regions with a larger edit bound align their metrics one pair at a time, the wave on one pair (wfa_ed_wave) */
enum { WD_DEFER = -1 };      /* internal: not this kernel's region after all */
enum { WD_SKIP = -2 };       /* internal: not this kernel's region by its record (avk_wide_static_ok): another launch is taking it (WideArgs::skip_static) */
/* instrumented emulator builds (-DAVK_WIDE_STATS): how often each hand-over site of solve_wide fired (tools/wide_defer_stats.py) */
#ifdef AVK_WIDE_STATS
extern uint64_t g_wide_defer[64];
extern uint32_t g_wide_defer_region[64]; /* the caller's index of the last region handed over at each site */
#define AVK_WDEFER(k) ((wv_lane() == 0 ? (void)(g_wide_defer[k] += 1, g_wide_defer_region[k] = a.regions[r].orig) : (void)0), (int)WD_DEFER)
#elif defined(AVK_WIDE_TIMING)
/* (profiling builds: the sites in four groups — class limits 1-5, inexact nodes 6-7, capacities 8-12, later phases 13-16 — in the last four profiling words) */
#define AVK_WDEFER(k) (wt.t[12 + ((k) <= 5 ? 0 : (k) <= 7 ? 1 : (k) <= 12 ? 2 : 3)] += 1, (int)WD_DEFER)
#else
#define AVK_WDEFER(k) ((int)WD_DEFER)
#endif

/* profiling builds (-DAVK_WIDE_TIMING, make wide-timing): clock ticks per phase of solve_wide, summed per wave and added to the spare words of the partial
 * tally at the end of the launch (avk_debug_phase_cycles reads them after a download; tools/gpu_wide_phases.py):
 * [0] record + tables, [1] rounds (the expansions), [2] commits, [3] genotype searches, [4] per-call outputs + alignments, [5] groups + outputs,
 * [6] whole regions, [7] regions, [8] rounds, [9] committed pops, [10] entries expanded, [11] handed over */
#ifdef AVK_WIDE_TIMING
struct WTime {
    u64 t[16], last;
};
#define AVK_WT_MARK(k)                      \
    {                                       \
        const u64 n_ = avk_clock();         \
        wt.t[k] += n_ - wt.last;            \
        wt.last = n_;                       \
    }
#define AVK_WT_COUNT(k, n) wt.t[k] += (n)
#define AVK_WT_ARG , WTime &wt
#define AVK_WT_PASS , wt
#else
#define AVK_WT_MARK(k)
#define AVK_WT_COUNT(k, n)
#define AVK_WT_ARG
#define AVK_WT_PASS
#endif

/* The wave's LDS, in words: a fixed head, then the region's sequence table, then the working area — what is left of the launch's LDS:
 *   phasing search:    free node slots [pool] | free wavefront blocks [blocks] | nodes [pool][W_NODE_W] | wavefront blocks [blocks][block words]
 *   genotype searches: the optima's haplotype states [optima][8] | one queue of W_QB_W words per search
 *   metrics:           one alignment scratch per lane that aligns */
enum {
    WO_VW0 = 0,    /* [16] call slot: rel_pos | a0_len << 8 | a1_len << 16 | type << 24 | zyg << 28 */
    WO_VW1 = 16,   /* [16] alt_ed | raw_space << 8 */
    WO_A1LO = 32,  /* [16] allele1, 2 bits per base: bases 0-15 */
    WO_A1HI = 48,  /* [16] bases 16-31 */
    WO_ORD = 64,   /* [16] search depth d: slot of its call | sync point << 8 (order_variants, query_optimizer.rs:372-381) */
    WO_ORDV = 80,  /* [16] search depth d: the call's slot word (as WO_VW0) */
    WO_SEL = 96,   /* [16] the round's entries: node | depth << 8 */
    WO_RES = 112,  /* [16][4] the round's results: child 0, child 1 (node | cost << 8 | has wavefront << 24 | inexact << 25), kind */
    WO_OPT = 176,  /* [64] nodes of the tied optima (| has wavefront << 8), in the order the reference finds them */
    WO_FILT = 240, /* [12][2][2] per-type alignments of the metrics phase: x | z << 16 */
    WO_CNT = 288,  /* [0] free wavefront blocks */
    WO_META = 296, /* [NSEQ] sequence s: length | failed_ed << 16; then the sequences, W1 words each; then the working area */
};

struct WideArgs {
    u32 lds_words;   /* LDS words per wave of this launch */
    u32 skip_static; /* records that fail avk_wide_static_ok() are left alone (a launch with AvkKernelArgs::only_not_wide takes them) instead of handed over */
};

/* the context avk_lane.inl's primitives run on: tables shared by the wave, the wavefront rows private to the lane */
struct WCtx {
    u32 *lds;
    u32 *seq;  /* sequence s at seq + s * W1 */
    u32 *meta;
    u32 *wfp;  /* this lane's wavefront rows */
    u32 W1, L, T, Q, N, qbase;
    u32 wfcap, wfcap_c;
    u32 jw0;   /* the slot word of the call this lane's haplotype step applies */
    u32 coop;  /* metrics phase of a region with a large edit bound: wfa_ed does not align, it leaves a request for the whole wave (resolve_alignments) */
    mutable u32 req0, req1, req2, pending;
#ifdef AVK_LANE_SLOW_TILES
    mutable u32 n_pops, n_diag, n_words;
#endif
    enum { MV = WMV };
    AVK_DEV_M u32 *seq_word(u32 s, u32 k) const { return seq + s * W1 + k; }
    AVK_DEV_M u32 seq_stride() const { return 1u; }
    AVK_DEV_M u32 seq_at(u32 k) const { return k; }
    AVK_DEV_M u32 *wf_row(u32, u32 row) const { return wfp + row; }
    AVK_DEV_M u32 vw0_at(u32 slot) const { return lds[WO_VW0 + slot]; }
    AVK_DEV_M u32 vw1_at(u32 slot) const { return lds[WO_VW1 + slot]; }
    AVK_DEV_M u32 step_w0(u32) const { return jw0; }
    AVK_DEV_M u32 vw0_side(u32 side, u32 j) const { return lds[WO_VW0 + WMV * side + j]; }
    AVK_DEV_M u32 seq_id(u32 side, u32 mask) const { return mask == 0 ? 0u : (side ? qbase + mask : mask); } /* truth masks 1 .. 2^T - 1, then query masks */
    AVK_DEV_M u32 seq_len(u32 s) const { return meta[s] & 0xFFFFu; }
    AVK_DEV_M u32 seq_fail(u32 s) const { return meta[s] >> 16; }
};

/* A haplotype state in 4 words (every field fits: positions and lengths <= 255, skip penalties <= the region's edit bound <= 255).  Word 3 is the
 * wavefront of a state at distance 0 (the one offset d0) AND of a state at distance 1 (its three offsets, one byte each: the usual distance of a node
 * that is not exact); from distance 2 on the offsets are in a wavefront block, bits 24-31 of word 2 = 1 + the block. */
AVK_DEV void hap_unpack(Hap &h, const u32 *w) {
    const u32 w0 = w[0], w1 = w[1], w2 = w[2];
    h.t_refpos = w0 & 0xFFu, h.q_refpos = (w0 >> 8) & 0xFFu, h.t_len = (w0 >> 16) & 0xFFu, h.q_len = w0 >> 24;
    h.t_skip = w1 & 0xFFu, h.q_skip = (w1 >> 8) & 0xFFu, h.nskip = (w1 >> 16) & 0xFFu, h.ed = w1 >> 24;
    h.t_alt = w2 & 0xFFu, h.q_alt = (w2 >> 8) & 0xFFu, h.t_nal = (w2 >> 16) & 0xFu, h.q_nal = (w2 >> 20) & 0xFu;
    h.d0 = w[3];
}
AVK_DEV void hap_pack(u32 *w, const Hap &h, u32 blk1 = 0) {
    w[0] = h.t_refpos | (h.q_refpos << 8) | (h.t_len << 16) | (h.q_len << 24);
    w[1] = h.t_skip | (h.q_skip << 8) | (h.nskip << 16) | (h.ed << 24);
    w[2] = h.t_alt | (h.q_alt << 8) | (h.t_nal << 16) | (h.q_nal << 20) | (blk1 << 24);
    if (h.ed == 0) w[3] = h.d0; /* (distance 1: the aligner wrote its front there) */
}

/* wfa_ed (src/util/sequence_alignment.rs:9-13) of sequences sa (baseline) and sb by the WHOLE WAVE: diagonal i of the front in lane i & 63, register i >> 6
 * (distances up to 63), every lane slides its diagonals (dynamic_wfa.rs:94-130), one ballot says whether a diagonal is full (:237-245), the next front
 * (:140-173, no clipping) is two moves along the lanes.  A lane on its own walks the 2 ed + 1 diagonals of every front one after the other — 25 ed^2
 * instructions, half a millisecond at distance 50; this is 50 ed.  Returns the distance, or -1 beyond 63. */
AVK_DEV int wfa_ed_wave(const WCtx &c, u32 sa, u32 la, u32 sb, u32 lb) {
    const u32 lane = (u32)wv_lane();
    u32 w0 = 0, w1 = 0, ed = 0;
    for (;;) {
        const u32 nd = 2 * ed + 1;
        bool full = false;
        if (lane < nd) {
            w0 += lane::match_run(c, sa, w0 + ed - lane, la, sb, w0, lb);
            full = w0 + ed - lane >= la && w0 >= lb;
        }
        if (nd > 64u && 64u + lane < nd) {
            w1 += lane::match_run(c, sa, w1 + ed - (64u + lane), la, sb, w1, lb);
            full = full || (w1 + ed - (64u + lane) >= la && w1 >= lb);
        }
        if (wv_ballot(full) != 0) return (int)ed;
        if (ed >= 63u) return -1;
        /* new[k] = max(old[k], old[k - 1] + 1, old[k - 2] + 1) over the entries that exist (old: k < nd) */
        const u32 a0 = wv_from_below(w0), top0 = wv_readlane(w0, 63);
        u32 a1 = wv_from_below(w1);
        a1 = lane == 0 ? top0 : a1; /* old[k - 1]: entry 64 takes entry 63 */
        const u32 b0 = wv_from_below(a0), top1 = wv_readlane(a0, 63);
        u32 b1 = wv_from_below(a1);
        b1 = lane == 0 ? top1 : b1; /* old[k - 2] */
        {
            const u32 k = lane;
            u32 v = k < nd ? w0 : 0u;
            if (k >= 1 && k - 1 < nd) v = a0 + 1 > v ? a0 + 1 : v;
            if (k >= 2 && k - 2 < nd) v = b0 + 1 > v ? b0 + 1 : v;
            w0 = v;
        }
        {
            const u32 k = 64u + lane;
            u32 v = k < nd ? w1 : 0u;
            if (k - 1 < nd) v = a1 + 1 > v ? a1 + 1 : v;
            if (k - 2 < nd) v = b1 + 1 > v ? b1 + 1 : v;
            w1 = v;
        }
        ed += 1;
    }
}
/* DWFALite::update / finalize of ONE haplotype state by the whole wave, from the front a lane left in its wavefront block (distance `ed`, offsets as bytes;
 * the lane stopped at W_COOP_ED, or did not start because the state was past it already): extend, then raise the distance until an end is touched
 * (update, dynamic_wfa.rs:68-84) or — `fin` — until a diagonal is full (update, then finalize :183-198: the same sequence of fronts, the first full one ends
 * both).  Diagonal i in lane i & 63, register i >> 6, as wfa_ed_wave.  The front goes back to the block.  Returns the distance, or -1 when the block's
 * `cap` entries do not hold the next front (the distance is more than `ed_out`). */
AVK_DEV int dw_continue_wave(const WCtx &c, u32 *blk, u32 ed, u32 cap, u32 sa, u32 la, u32 sb, u32 lb, bool fin, u32 &ed_out) {
    const u32 lane = (u32)wv_lane();
    const u8 *bytes = (const u8 *)blk;
    u32 w0 = lane < 2 * ed + 1 ? bytes[lane] : 0u, w1 = 64u + lane < 2 * ed + 1 ? bytes[64u + lane] : 0u;
    int ret;
    for (;;) {
        const u32 nd = 2 * ed + 1;
        bool stop = false;
        if (lane < nd) {
            w0 += lane::match_run(c, sa, w0 + ed - lane, la, sb, w0, lb);
            const bool eb = w0 + ed - lane >= la, eo = w0 >= lb;
            stop = fin ? (eb && eo) : (eb || eo);
        }
        if (nd > 64u && 64u + lane < nd) {
            w1 += lane::match_run(c, sa, w1 + ed - (64u + lane), la, sb, w1, lb);
            const bool eb = w1 + ed - (64u + lane) >= la, eo = w1 >= lb;
            stop = stop || (fin ? (eb && eo) : (eb || eo));
        }
        if (wv_ballot(stop) != 0) {
            ret = (int)ed;
            break;
        }
        if (2 * ed + 3 > cap) {
            ret = -1;
            break;
        }
        const u32 a0 = wv_from_below(w0), top0 = wv_readlane(w0, 63);
        u32 a1 = wv_from_below(w1);
        a1 = lane == 0 ? top0 : a1;
        const u32 b0 = wv_from_below(a0), top1 = wv_readlane(a0, 63);
        u32 b1 = wv_from_below(a1);
        b1 = lane == 0 ? top1 : b1;
        {
            const u32 k = lane;
            u32 v = k < nd ? w0 : 0u;
            if (k >= 1 && k - 1 < nd) v = a0 + 1 > v ? a0 + 1 : v;
            if (k >= 2 && k - 2 < nd) v = b0 + 1 > v ? b0 + 1 : v;
            w0 = v;
        }
        {
            const u32 k = 64u + lane;
            u32 v = k < nd ? w1 : 0u;
            if (k - 1 < nd) v = a1 + 1 > v ? a1 + 1 : v;
            if (k - 2 < nd) v = b1 + 1 > v ? b1 + 1 : v;
            w1 = v;
        }
        ed += 1;
    }
    wv_sync();
    u8 *out = (u8 *)blk;
    if (lane < 2 * ed + 1) out[lane] = (u8)w0;
    if (64u + lane < 2 * ed + 1) out[64u + lane] = (u8)w1;
    wv_sync();
    ed_out = ed;
    return ret;
}

/* What avk_lane.inl's distance functions call when they have to align (found by argument-dependent lookup: this context is of this namespace).  Small
 * edit bounds: the lane aligns by itself, side by side with the others.  Large ones (WCtx::coop): the lane leaves its request, the wave takes the requests
 * one after the other (resolve_alignments). */
AVK_DEV int wfa_ed(const WCtx &c, u32 sa, u32 la, u32 sb, u32 lb) {
    if (c.coop) {
        c.req0 = sa | (la << 16);
        c.req1 = sb | (lb << 16);
        c.pending = 1;
        return 0;
    }
    return lane::wfa_ed<WCtx>(c, sa, la, sb, lb);
}
/* the requests the lanes left in wfa_ed: each is aligned by the whole wave, its lane gets the distance in `val`; false: a distance beyond wfa_ed_wave's reach */
AVK_DEV bool resolve_alignments(const WCtx &c, int &val) {
    const u32 lane = (u32)wv_lane();
    bool ok = true;
    for (u64 m = wv_ballot(c.pending != 0); m; m &= m - 1) {
        const u32 src = (u32)avk_ctz64(m);
        const u32 r0 = wv_readlane(c.req0, src), r1 = wv_readlane(c.req1, src);
        const int e = wfa_ed_wave(c, r0 & 0xFFFFu, r0 >> 16, r1 & 0xFFFFu, r1 >> 16);
        ok = ok && e >= 0;
        if (lane == src) val = e;
    }
    c.pending = 0;
    return ok;
}

/* FULL(side, mask): the calls of the mask applied in the side's order; a call that starts before the end of the previous applied one is dropped and
 * its alt_ed counted (generate_allele_sequence, waffle_solver.rs:726-778; build_full of avk_lane.inl with the records in LDS) */
AVK_DEV void build_full(const WCtx &c, u32 side, u32 mask) {
    const u32 s = c.seq_id(side, mask);
    lane::SeqWriter w;
    w.acc = 0;
    w.nb = 0;
    w.row = s;
    w.k = 0;
    u32 cur = 0, len = 0, failed = 0;
    for (u32 left = mask; left; left &= left - 1) {
        const u32 slot = WMV * side + (u32)__builtin_ctz(left);
        const u32 w0 = c.lds[WO_VW0 + slot];
        const u32 pos = w0 & 0xFFu, a0 = (w0 >> 8) & 0xFFu, a1 = (w0 >> 16) & 0xFFu;
        if (pos < cur) {
            failed += c.lds[WO_VW1 + slot] & 0xFFu;
            continue;
        }
        lane::sw_ref(c, w, cur, pos);
        lane::sw_push(c, w, c.lds[WO_A1LO + slot], a1 < 16 ? a1 : 16u);
        if (a1 > 16) lane::sw_push(c, w, c.lds[WO_A1HI + slot], a1 - 16);
        len += pos - cur + a1;
        cur = pos + a0;
    }
    if (cur < c.L) {
        lane::sw_ref(c, w, cur, c.L);
        len += c.L - cur;
    }
    if (w.nb) *c.seq_word(w.row, w.k) = (u32)w.acc;
    c.meta[s] = len | (failed << 16);
}

/* ---- the sorted queue of the phasing search: entry j in lane j, ascending (cost, id) ------------------------------------------------------- */
struct WQ {
    u32 key[W_QR];  /* cost << 16 | id: the reference's NodePriority (Reverse(cost), Reverse(id)), query_optimizer.rs:470-475 */
    u32 info[W_QR]; /* node | depth << 8 | flags */
    u32 c0[W_QR], c1[W_QR]; /* expansion: child (or the finished node itself) as node | cost << 8 */
};
enum {
    QF_EXP = 1u << 13,     /* c0 / c1 hold the node's expansion */
    QF_FINAL = 1u << 14,   /* depth N: c0 is the finalised node (itself) with its final cost */
    QF_TWO = 1u << 15,     /* two cloned children (REF|ALT), (ALT|REF), :269-293; else one moved child that keeps the id, :294-327 */
    QF_INEXACT = 1u << 17, /* the node's own cost is a lower bound (a wavefront past W_ED_MAX): it cannot be expanded here */
    QF_C0X = 1u << 18,     /* child 0 / the final cost is inexact */
    QF_C1X = 1u << 19,
    QF_WF = 1u << 20,      /* a haplotype state of the node has a wavefront block */
};
/* c0 / c1 and the round's result words: node | cost << 8 | has wavefront << 24 | inexact << 25 */
enum { CH_WF = 1u << 24, CH_X = 1u << 25 };
/* entry i + 1 becomes entry i (rows that hold nothing are left alone: qn is wave-uniform) */
AVK_DEV void q_pop_front(WQ &q, u32 &qn) {
    const u32 lane = (u32)wv_lane();
#pragma unroll
    for (int r = 0; r < W_QR; ++r) {
        if ((u32)r * 64u >= qn) break;
        u32 k = wv_from_above(q.key[r]), i = wv_from_above(q.info[r]), a = wv_from_above(q.c0[r]), b = wv_from_above(q.c1[r]);
        if (r + 1 < W_QR && (u32)(r + 1) * 64u < qn) { /* lane 63 takes the next row's first entry (not moved yet) */
            const u32 nk = wv_readlane(q.key[r + 1 < W_QR ? r + 1 : r], 0), ni = wv_readlane(q.info[r + 1 < W_QR ? r + 1 : r], 0);
            const u32 na = wv_readlane(q.c0[r + 1 < W_QR ? r + 1 : r], 0), nb = wv_readlane(q.c1[r + 1 < W_QR ? r + 1 : r], 0);
            if (lane == 63) k = nk, i = ni, a = na, b = nb;
        }
        q.key[r] = k, q.info[r] = i, q.c0[r] = a, q.c1[r] = b; /* (lanes from the last entry on hold leftovers: never read) */
    }
    qn -= 1;
}
/* the caller has checked qn < 64 W_QR */
AVK_DEV void q_insert(WQ &q, u32 &qn, u32 key, u32 info) {
    const u32 lane = (u32)wv_lane();
    u32 pos = 0;
#pragma unroll
    for (int r = 0; r < W_QR; ++r)
        if ((u32)r * 64u < qn) pos += (u32)avk_popc64(wv_ballot((u32)r * 64u + lane < qn && q.key[r] < key));
#pragma unroll
    for (int r = W_QR - 1; r >= 0; --r) { /* entry i becomes entry i + 1 from `pos` on: the last row first, the row below is still in place */
        if ((u32)r * 64u > qn || (u32)r * 64u + 63u < pos) continue;
        u32 k = wv_from_below(q.key[r]), i = wv_from_below(q.info[r]), a = wv_from_below(q.c0[r]), b = wv_from_below(q.c1[r]);
        if (r > 0) {
            const u32 pk = wv_readlane(q.key[r > 0 ? r - 1 : 0], 63), pi = wv_readlane(q.info[r > 0 ? r - 1 : 0], 63);
            const u32 pa = wv_readlane(q.c0[r > 0 ? r - 1 : 0], 63), pb = wv_readlane(q.c1[r > 0 ? r - 1 : 0], 63);
            if (lane == 0) k = pk, i = pi, a = pa, b = pb;
        }
        const u32 gi = (u32)r * 64u + lane;
        if (gi > pos) q.key[r] = k, q.info[r] = i, q.c0[r] = a, q.c1[r] = b;
        if (gi == pos) q.key[r] = key, q.info[r] = info, q.c0[r] = 0, q.c1[r] = 0;
    }
    qn += 1;
}

/* ---- optimize_gt_alleles for one haplotype, one search per lane (exact_gt_optimizer.rs:108-357; phaseB of avk_lane.inl with the nodes' states stored
 * instead of replayed).  The aligner of an ExactMatchNode has max_edit_distance 0 (:380): a live node is exact, its wavefront the end of its
 * shorter sequence.  Entry i of the search's queue = words 4 i .. 4 i + 3 of `qb`: key, then the state's first three words.
 * key = errors << 26 | (31 - (depth - errors)) << 21 | id  (Reverse(errors), set - errors, Reverse(id); :452-458). */
AVK_DEV u32 keyB(u32 errors, u32 depth, u32 id) { return (errors << 26) | ((31u - (depth - errors)) << 21) | id; }
AVK_DEV bool hapB_step(const WCtx &c, Hap &h, u32 slot, u32 sync, bool alt) {
    const bool ok = lane::hap_step(c, h, slot < WMV, true, slot, alt ? L_ALT : L_REF, sync); /* (c.jw0 is the call's slot word) */
    const u32 st = c.seq_id(0, h.t_alt), sq = c.seq_id(1, h.q_alt);
    h.d0 += lane::match_run_same(c, st, h.t_len, sq, h.q_len, h.d0);
    const u32 lim = h.t_len < h.q_len ? h.t_len : h.q_len;
    return ok && h.d0 >= lim;
}
AVK_DEV int pushB(u32 *qb, u32 &qn, u32 key, const Hap &h) {
    if (qn >= (u32)W_QB) return WD_DEFER;
    u32 w[4];
    hap_pack(w, h);
    u32 *e = qb + 4u * qn;
    e[0] = key, e[1] = w[0], e[2] = w[1], e[3] = w[2];
    qn += 1;
    return 0;
}
/* returns the number of flips and the final alleles, WD_DEFER, or -100 - status */
AVK_DEV int phaseB(WCtx &c, u32 *qb, u32 in_t, u32 in_q, u32 &res_t, u32 &res_q) {
    u32 qn = 0;
    {
        Hap root;
        lane::hap_init(root);
        pushB(qb, qn, keyB(0, 0, 0), root);
    }
    u32 next_id = 1, min_sync = 0, af_counts = 0;
    while (qn > 0) {
        u32 best = 0xFFFFFFFFu, bi = 0;
        for (u32 i = 0; i < qn; ++i) {
            const u32 k = qb[4u * i];
            if (k < best) best = k, bi = i;
        }
        Hap h;
        {
            u32 *e = qb + 4u * bi, *l = qb + 4u * (qn - 1);
            const u32 w[4] = {e[1], e[2], e[3], 0u};
            hap_unpack(h, w);
            e[0] = l[0], e[1] = l[1], e[2] = l[2], e[3] = l[3];
            qn -= 1;
        }
        const u32 errors = best >> 26, id = best & 0x1FFFFFu;
        h.d0 = h.t_len < h.q_len ? h.t_len : h.q_len; /* every queued node is exact */
        const u32 depth = h.t_nal + h.q_nal;
        if (depth == c.N) { /* :180-192 */
            lane::hap_step(c, h, true, false, 0, L_REF, c.L);
            const u32 st = c.seq_id(0, h.t_alt), sq = c.seq_id(1, h.q_alt);
            h.d0 += lane::match_run_same(c, st, h.t_len, sq, h.q_len, h.d0);
            if (h.d0 >= h.t_len && h.d0 >= h.q_len) {
                res_t = h.t_alt;
                res_q = h.q_alt;
                return (int)errors;
            }
            continue;
        }
        if (depth < min_sync) continue; /* :194-197 */
        if (h.t_len == h.q_len && h.t_refpos == h.q_refpos) { /* :206-217 */
            min_sync = depth;
            af_counts = 0;
        }
        const u32 o = c.lds[WO_ORD + depth], slot = o & 0xFFu, sync = o >> 8;
        c.jw0 = c.lds[WO_ORDV + depth];
        const bool cur_alt = (((slot < WMV ? in_t : in_q) >> (slot & 7u)) & 1u) != 0;
        if (!cur_alt) { /* :257-273 */
            if (hapB_step(c, h, slot, sync, false)) {
                if (pushB(qb, qn, keyB(errors, depth + 1, id), h)) return WD_DEFER;
            }
        } else { /* :274-306: (REF, error) first, then ALT */
            Hap r = h;
            if (hapB_step(c, r, slot, sync, false)) {
                if (pushB(qb, qn, keyB(errors + 1, depth + 1, next_id), r)) return WD_DEFER;
            }
            next_id += 1;
            if (hapB_step(c, h, slot, sync, true)) {
                if (pushB(qb, qn, keyB(errors, depth + 1, next_id), h)) return WD_DEFER;
            }
            next_id += 1;
        }
        af_counts += 1;
        if (af_counts >= 500u) return WD_DEFER; /* the auto-fail pruning (:309-339) is the wave-per-region kernel's */
    }
    return -100 - AVK_ST_NO_GT_RESULT; /* :345-348 */
}

/* ---- one region ---------------------------------------------------------------------------------------------------------------------------- */
/* DWFALite::update (and finalize, `fin`) of the haplotype state H, read from srcp and about to be written to dstp (the same place when a finished node is
 * finalised).  Where the wavefront lives depends on the distance: at 0 it is H.d0 and the state is first advanced without memory (most stay at 0); at 1
 * its three offsets are the state's fourth word, the aligner works on dstp[3]; from 2 on it takes a block of the pool (the parent's is copied; a state
 * updated in place keeps its own).  db1 = 1 + the block of the result (0: none).  Returns 0, or nonzero when the state could not be brought up to date
 * (front of W_ED_MAX full, or no block left): the distance is then at least H.ed + more (more: 0 or 1). */
AVK_DEV int hap_advance(const WCtx &c, u32 *blocks, u32 blk_w, u32 *wfree, Hap &H, const u32 *srcp, u32 *dstp, bool fin, u32 &db1, u32 &more) {
    const int LS_PARTIAL_ = (int)lane::LS_PARTIAL;
    WCtx cj = c;
    int rr = 0;
    db1 = 0;
    more = 0;
    const u32 sb1 = H.ed >= 2 ? (srcp[2] >> 24) : 0u;
    if (H.ed == 0) {
        rr = lane::hap_update(cj, H, 0, 0u); /* 0, or LS_PARTIAL: a mismatch with both sequences going on */
        if (rr == 0 && fin) rr = lane::hap_finalize(cj, H, 0, 0u);
        if (rr == 0) return 0;
    }
    if (H.ed <= 1) { /* the front fits the state's fourth word */
        if (H.ed == 1 && dstp != srcp) dstp[3] = srcp[3];
        cj.wfp = dstp + 3;
        cj.wfcap = 4;
        rr = lane::hap_update(cj, H, 0);
        if (!rr && fin) rr = lane::hap_finalize(cj, H, 0);
        if (rr == 0) return 0;
        /* the front of distance 1 is extended as far as it goes and touches no end: distance 2 and more */
    }
    if (dstp == srcp && sb1) db1 = sb1;
    else {
        const u32 old = avk_wg_add(c.lds + WO_CNT, 0xFFFFFFFFu);
        if ((int)old <= 0) { /* no block left */
            avk_wg_add(c.lds + WO_CNT, 1u);
            more = H.ed <= 1 ? 1u : 0u;
            return 1;
        }
        db1 = wfree[old - 1u] + 1u;
        u32 *dp = blocks + (db1 - 1u) * blk_w;
        if (sb1) {
            const u32 *sp = blocks + (sb1 - 1u) * blk_w;
            for (u32 k = 0; k < (2 * H.ed + 1 + 3) >> 2; ++k) dp[k] = sp[k];
        } else dp[0] = dstp[3];
    }
    cj.wfp = blocks + (db1 - 1u) * blk_w;
    cj.wfcap = c.wfcap;
    /* a lane follows the front up to distance W_COOP_ED; from there every new front is the whole wave's (a lane walks the 2 ed + 1 diagonals of each front
     * one after the other: a state that meets a 50-base difference took half a millisecond) */
    rr = LS_PARTIAL_;
    if (H.ed < (u32)W_COOP_ED) {
        rr = lane::hap_update(cj, H, 0, (u32)W_COOP_ED);
        if (!rr && fin) rr = lane::hap_finalize(cj, H, 0, (u32)W_COOP_ED);
    }
    if (rr == LS_PARTIAL_) { /* the request (resolve_updates) */
        const u32 st = c.seq_id(0, H.t_alt), sq = c.seq_id(1, H.q_alt);
        c.req0 = st | (H.t_len << 16);
        c.req1 = sq | (H.q_len << 16);
        c.req2 = (db1 - 1u) | (H.ed << 8) | (fin ? 0x10000u : 0u);
        c.pending = 1;
        return 0;
    }
    more = rr ? 1u : 0u; /* the front at H.ed is full and touches no end: the distance is more */
    return rr;
}
/* the requests the lanes of a round left in hap_advance: the whole wave takes each state's aligner on, the state's lane gets the distance */
AVK_DEV void resolve_updates(const WCtx &c, u32 *blocks, u32 blk_w, Hap &H, int &rr, u32 &more) {
    const u32 lane = (u32)wv_lane();
    for (u64 m = wv_ballot(c.pending != 0); m; m &= m - 1) {
        const u32 src = (u32)avk_ctz64(m);
        const u32 r0 = wv_readlane(c.req0, src), r1 = wv_readlane(c.req1, src), r2 = wv_readlane(c.req2, src);
        u32 ed_out = 0;
        const int e = dw_continue_wave(c, blocks + (r2 & 0xFFu) * blk_w, (r2 >> 8) & 0xFFu, c.wfcap, r0 & 0xFFFFu, r0 >> 16, r1 & 0xFFFFu, r1 >> 16, (r2 >> 16) != 0, ed_out);
        if (lane == src) {
            H.ed = ed_out;
            rr = e < 0 ? 1 : 0;
            more = e < 0 ? 1u : 0u;
        }
    }
    c.pending = 0;
}

/* returns AVK_ST_* (>= 0) or WD_DEFER */
AVK_DEV int solve_wide(const AvkKernelArgs &a, const WideArgs &wa, u32 r, u32 *lds, u64 *part, lane::LaneOut &out AVK_WT_ARG) {
    const u32 lane = (u32)wv_lane();
    const u64 below = lane ? (~0ull >> (64u - lane)) : 0ull;
    const AvkDevRegion reg = a.regions[r];
    const u32 L = wv_uni(reg.len), T = wv_uni(reg.t_cnt), Q = wv_uni(reg.q_cnt), N = T + Q;
    const u32 grow = wv_uni(reg.grow), ed_bound = wv_uni(reg.ed_bound);
    const u32 orig = wv_uni(reg.orig), v_off = wv_uni(reg.v_off), pre = wv_uni(reg.pre_status);
    /* at most WMV calls on a side; every offset of every wavefront is a byte, and an offset can pass its sequence's end by the distance
     * (increase_edit_distance does not clip) */
    if (!avk_wide_static_ok(L, grow, ed_bound, T, Q, pre)) return wa.skip_static ? (int)WD_SKIP : AVK_WDEFER(1);
    WCtx c;
    c.lds = lds;
    c.L = L, c.T = T, c.Q = Q, c.N = N;
    c.W1 = ((L + grow + 15u) >> 4) + 1u;
    c.qbase = (1u << T) - 1u;
    c.jw0 = 0;
    c.coop = 0, c.req0 = 0, c.req1 = 0, c.pending = 0;
    const u32 nseq = (1u << T) + (1u << Q) - 1u;
    c.meta = lds + WO_META;
    c.seq = c.meta + nseq;
    u32 *const dyn = c.seq + nseq * c.W1;
    const u64 fixed_words = (u64)WO_META + (u64)nseq * (c.W1 + 1u);
    const u32 ed_node = ed_bound < 1u ? 1u : (ed_bound < (u32)W_ED_MAX ? ed_bound : (u32)W_ED_MAX);
    const u32 blk_w = ((2 * ed_node + 2 + 3) >> 2) | 1u; /* words of a wavefront block (odd: spreads the banks) */
    if (fixed_words + (u64)W_NBLK_MIN * (blk_w + 1u) + (u64)W_POOL_MIN * (W_NODE_W + 1u) > (u64)wa.lds_words) return AVK_WDEFER(3);
    const u32 dyn_words = wa.lds_words - (u32)fixed_words;
    c.wfcap = 2 * ed_node + 2;
    const u32 scr_w = (2 * ed_bound + 3 + 3) >> 2; /* words of an alignment scratch of the metrics phase: no two strings of the region are further apart than the bound */
    c.wfcap_c = 4 * scr_w;
    c.wfp = dyn;

    /* ---- the region's tables */
    { /* reference window: 2 bits per base from the packed genome; a flagged word (anything but upper-case ACGT) is not for this kernel */
        const u64 w0 = reg.ref_off >> 4;
        const u32 shift = (u32)(reg.ref_off & 15u), nw = (L + shift + 15u) >> 4;
        bool exc = false;
        for (u32 k = lane; k < nw; k += 64) {
            const u64 w = w0 + k;
            exc = exc || ((a.ref_exc[w >> 5] >> (w & 31)) & 1u);
        }
        if (wv_ballot(exc) != 0) return AVK_WDEFER(4);
        for (u32 k = lane; k < c.W1; k += 64) {
            u32 word = 0;
            if (k * 16 < L) {
                const u32 lo = a.ref_2bit[w0 + k], hi = a.ref_2bit[w0 + k + 1];
                word = (u32)((((u64)hi << 32) | lo) >> (2 * shift));
            }
            c.seq[k] = word;
        }
        if (lane == 0) c.meta[0] = L;
    }
    const u32 types = pre >> 16;
    { /* the blob (AvkBlobVar / AvkOrdVar, avk_dev_types.h): lane s takes call slot s, lane d search depth d */
        const u32 *blob = a.blob + 2ull * wv_uni(reg.blob_off);
        const AvkBlobVar *bv = (const AvkBlobVar *)blob;
        const u8 *ba = (const u8 *)blob + (((u64)N * sizeof(AvkBlobVar) + 15) & ~15ull);
        const AvkOrdVar *bo = (const AvkOrdVar *)(ba + (((u64)wv_uni(reg.alle_bytes) + 15) & ~15ull));
        bool bad = false;
        if (lane < (u32)WNS) {
            const u32 side = lane >> 3, j = lane & 7u;
            u32 w0 = 0, w1 = 0, lo = 0, hi = 0;
            if (j < (side ? Q : T)) {
                const AvkBlobVar v = bv[side ? T + j : j];
                const u32 vt = v.type_zyg & 0xFFu, zy = (v.type_zyg >> 8) & 0xFFu;
                bad = v.rel_pos > 255u || v.a0_len > 255u || v.a1_len > 32u || v.alt_ed > 255u || v.raw_space > 0xFFFFu || vt > 15u || zy > 7u;
                const u8 *a1 = ba + v.a_off + v.a0_len;
                for (u32 b = 0; b < v.a1_len && !bad; ++b) {
                    const u8 ch = a1[b];
                    const u32 code = ch == 'C' ? 1u : (ch == 'G' ? 2u : (ch == 'T' ? 3u : 0u));
                    bad = !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
                    if (b < 16) lo |= code << (2 * b);
                    else hi |= code << (2 * (b - 16));
                }
                w0 = v.rel_pos | (v.a0_len << 8) | (v.a1_len << 16) | (vt << 24) | (zy << 28);
                w1 = v.alt_ed | (v.raw_space << 8);
            }
            lds[WO_VW0 + lane] = w0, lds[WO_VW1 + lane] = w1, lds[WO_A1LO + lane] = lo, lds[WO_A1HI + lane] = hi;
            if (lane < N) {
                const AvkOrdVar o = bo[lane];
                lds[WO_ORD + lane] = (o.vi < T ? o.vi : (u32)WMV + o.vi - T) | (o.sync << 8);
                lds[WO_ORDV + lane] = (o.rel_pos & 0xFFu) | ((o.a0_len & 0xFFu) << 8) | ((o.a1_len & 0xFFu) << 16) | ((o.type_zyg & 0xFu) << 24) | (((o.type_zyg >> 8) & 7u) << 28);
            }
        }
        if (wv_ballot(bad) != 0) return AVK_WDEFER(5);
    }
    wv_sync();
    for (u32 s = 1 + lane; s < nseq; s += 64) { /* the full-length sequences, one per lane at a time */
        const u32 side = s > c.qbase ? 1u : 0u;
        build_full(c, side, side ? s - c.qbase : s);
    }

    AVK_WT_MARK(0)
    /* ---- phase A: optimize_sequences */
    /* seven tenths of the working area for nodes (at most W_POOL_MAX), the rest for wavefront blocks */
    const u32 pool_want = (dyn_words * 7u / 10u) / (W_NODE_W + 1u);
    const u32 pool_n = pool_want < (u32)W_POOL_MAX ? pool_want : (u32)W_POOL_MAX;
    const u32 blk_room = (dyn_words - pool_n * (W_NODE_W + 1u)) / (blk_w + 1u);
    const u32 n_blk = blk_room < (u32)W_NBLK_MAX ? blk_room : (u32)W_NBLK_MAX;
    u32 *const nfree_list = dyn, *const wfree = dyn + pool_n, *const nodes = wfree + n_blk, *const blocks = nodes + pool_n * W_NODE_W;
    u32 nfree = pool_n - 1; /* node 0 is the root */
    for (u32 k = lane; k < nfree; k += 64) nfree_list[k] = pool_n - 1 - k; /* (the stack's top is the lowest slot) */
    for (u32 k = lane; k < n_blk; k += 64) wfree[k] = k;
    if (lane < (u32)W_NODE_W) nodes[lane] = 0; /* the root: two empty haplotypes */
    if (lane == 0) lds[WO_CNT] = n_blk;
    WQ q;
#pragma unroll
    for (int rr = 0; rr < W_QR; ++rr) q.key[rr] = 0, q.info[rr] = 0, q.c0[rr] = 0, q.c1[rr] = 0; /* entry 0: the root (cost 0, id 0, node 0, depth 0) */
    u32 qn = 1, next_id = 1, best = 0xFFFFFFFFu, nbest = 0;
    u32 rbucket = 0; /* lane d: nodes of depth d looked at (the per-depth quota, :222-225) */
    const u32 max_branch = a.max_branch_factor;
    /* a node goes back to the pool — with the wavefront blocks of its states, when it has any (rare: the flag spares the look) */
    auto free_node = [&](u32 slot_wf) {
        if (lane == 0) {
            const u32 slot = slot_wf & 0xFFu;
            nfree_list[nfree] = slot;
            if (slot_wf & CH_WF) {
                u32 cnt = lds[WO_CNT];
                const u32 b0 = nodes[slot * W_NODE_W + 2] >> 24, b1 = nodes[slot * W_NODE_W + 6] >> 24;
                if (b0) wfree[cnt++] = b0 - 1u;
                if (b1) wfree[cnt++] = b1 - 1u;
                lds[WO_CNT] = cnt;
            }
        }
        nfree += 1;
    };
    for (;;) {
        bool need_round = false;
        while (qn > 0) {
            const u32 key0 = wv_readlane(q.key[0], 0), info0 = wv_readlane(q.info[0], 0), c0 = wv_readlane(q.c0[0], 0), c1 = wv_readlane(q.c1[0], 0);
            const u32 cost = key0 >> 16;
            if (cost > best) { /* :204 skips it — and, pops being in non-decreasing cost order, everything behind it */
                qn = 0;
                break;
            }
            const u32 depth = (info0 >> 8) & 31u, node0 = (info0 & 0xFFu) | ((info0 & QF_WF) ? (u32)CH_WF : 0u);
            if (wv_readlane(rbucket, depth) >= max_branch) { /* :222: dropped — with what was made of it ahead of its turn */
                if ((info0 & QF_EXP) && !(info0 & QF_FINAL)) {
                    free_node(c0);
                    if (info0 & QF_TWO) free_node(c1);
                }
                free_node((info0 & QF_FINAL) && (info0 & QF_EXP) ? (c0 & (0xFFu | CH_WF)) : node0);
                q_pop_front(q, qn);
                continue;
            }
            if (info0 & QF_INEXACT) return AVK_WDEFER(6); /* its turn has come and its state is not known */
            if (!(info0 & QF_EXP)) {
                need_round = true;
                break;
            }
            if (lane == depth) rbucket += 1;
            AVK_WT_COUNT(9, 1);
            q_pop_front(q, qn);
            if (info0 & QF_FINAL) { /* :227-247 */
                if (c0 & CH_X) return AVK_WDEFER(7);
                const u32 fc = (c0 >> 8) & 0xFFFFu;
                if (fc < best) {
                    for (u32 k = 0; k < nbest; ++k) {
                        const u32 o = wv_uni(lds[WO_OPT + k]);
                        free_node((o & 0xFFu) | ((o >> 8) ? (u32)CH_WF : 0u));
                    }
                    best = fc;
                    nbest = 0;
                }
                if (fc == best) {
                    if (nbest >= (u32)W_OPTCAP) return AVK_WDEFER(8);
                    if (lane == 0) lds[WO_OPT + nbest] = (c0 & 0xFFu) | ((c0 & CH_WF) ? 0x100u : 0u);
                    nbest += 1;
                } else free_node(c0);
                continue;
            }
            /* the children, in the reference's order: (REF|ALT) then (ALT|REF) with new ids, or the moved node with its own */
            const u32 n_child = (info0 & QF_TWO) ? 2u : 1u;
            if (qn + n_child > 64u * (u32)W_QR) return AVK_WDEFER(9);
            for (u32 k = 0; k < n_child; ++k) {
                const u32 ch = k ? c1 : c0;
                const u32 id = (info0 & QF_TWO) ? next_id++ : (key0 & 0xFFFFu);
                q_insert(q, qn, (((ch >> 8) & 0xFFFFu) << 16) | id, (ch & 0xFFu) | ((depth + 1u) << 8) | ((ch & CH_X) ? (u32)QF_INEXACT : 0u) | ((ch & CH_WF) ? (u32)QF_WF : 0u));
            }
            free_node(node0);
            if (next_id > 60000u) return AVK_WDEFER(11);
        }
        AVK_WT_MARK(2)
        if (!need_round) break;

        /* ---- a round: the first entries that have no expansion yet, all at once */
        wv_sync();
        const u32 mydepth = (q.info[0] >> 8) & 31u; /* (the first 64 entries are looked at) */
        const u32 mycnt = wv_shfl(rbucket, (int)mydepth);
        const bool cand = lane < qn && !(q.info[0] & (QF_EXP | QF_INEXACT)) && (q.key[0] >> 16) <= best && mycnt < max_branch;
        const u64 cm = wv_ballot(cand);
        u32 K = (u32)avk_popc64(cm);
        K = K < (u32)W_K ? K : (u32)W_K;
        K = K < nfree / 2 ? K : nfree / 2;
        if (K == 0) return AVK_WDEFER(12); /* the pool is exhausted */
        AVK_WT_COUNT(8, 1);
        AVK_WT_COUNT(10, K);
        const u32 rank = (u32)avk_popc64(cm & below);
        const bool sel = cand && rank < K;
        if (sel) lds[WO_SEL + rank] = q.info[0] & (0x1FFFu | QF_WF);
        wv_sync();
        const u32 e = lane >> 2, cc = (lane >> 1) & 1u, hh = lane & 1u;
        const bool act = e < K;
        const u32 sinfo = act ? lds[WO_SEL + e] : 0u;
        const u32 ns = sinfo & 0xFFu, d = (sinfo >> 8) & 31u;
        const bool fin = act && d == N;
        u32 slot = 0, sync = 0, zyg = 0, jw0 = 0;
        if (act && !fin) {
            const u32 o = lds[WO_ORD + d];
            jw0 = lds[WO_ORDV + d];
            slot = o & 0xFFu;
            sync = o >> 8;
            zyg = jw0 >> 28;
        }
        const bool is_truth = slot < (u32)WMV;
        const bool het = zyg == AVK_ZYG_UNPHASED_HET || zyg == AVK_ZYG_PHASED_HET01 || zyg == AVK_ZYG_PHASED_HET10;
        const bool two = act && !fin && het && (!is_truth || zyg == AVK_ZYG_UNPHASED_HET); /* :269-293 */
        const bool job = act && (fin ? cc == 0 : (two || cc == 0));
        u32 allele = L_ALT; /* hom-alt: (ALT, ALT), :313-327 */
        if (two) allele = (cc ^ hh) ? L_ALT : L_REF; /* child 0 = (REF|ALT), child 1 = (ALT|REF) */
        else if (het) allele = ((zyg == AVK_ZYG_PHASED_HET01) == (hh == 1)) ? L_ALT : L_REF; /* 0|1: REF on haplotype 1, ALT on haplotype 2 */
        const bool wants = job && !fin && hh == 0;
        const u64 am = wv_ballot(wants);
        u32 cnode = wants ? nfree_list[nfree - 1u - (u32)avk_popc64(am & below)] : 0u;
        cnode = wv_shfl(cnode, (int)(lane & ~1u));
        nfree -= (u32)avk_popc64(am);
        const u32 dnode = fin ? ns : cnode;
        u32 mycost = 0, myflags = 0, db1 = 0, more = 0;
        int rr = 0;
        Hap H;
        lane::hap_init(H);
        u32 *dstp = nodes + dnode * W_NODE_W + 4 * hh;
        WCtx cj = c;
        cj.jw0 = jw0;
        cj.pending = 0;
        if (job) {
            const u32 *srcp = nodes + ns * W_NODE_W + 4 * hh;
            hap_unpack(H, srcp);
            if (fin) lane::hap_step(cj, H, true, false, 0, L_REF, L); /* ComparisonNode::finalize_dwfas (:457-462, haplotype_dwfa.rs:84-95) */
            else lane::hap_step(cj, H, is_truth, true, slot, allele, sync);
            rr = hap_advance(cj, blocks, blk_w, wfree, H, srcp, dstp, fin, db1, more);
        }
        resolve_updates(cj, blocks, blk_w, H, rr, more); /* (states whose distance passed W_COOP_ED: the whole wave, one state at a time) */
        if (job) {
            mycost = H.t_skip + H.q_skip + H.ed + more; /* (rr: a lower bound) */
            myflags = (rr ? (u32)CH_X : 0u) | (db1 ? (u32)CH_WF : 0u);
            hap_pack(dstp, H, db1);
        }
        const u32 ocost = wv_shfl(mycost, (int)(lane ^ 1u)), oflags = wv_shfl(myflags, (int)(lane ^ 1u));
        if (job && hh == 0) {
            const u32 cst = mycost + ocost;
            lds[WO_RES + 4 * e + cc] = dnode | ((cst < 0xFFFFu ? cst : 0xFFFFu) << 8) | myflags | oflags | (cst >= 0xFFFFu ? (u32)CH_X : 0u);
        }
        if (act && cc == 0 && hh == 0) lds[WO_RES + 4 * e + 2] = (fin ? (u32)QF_FINAL : 0u) | (two ? (u32)QF_TWO : 0u);
        wv_sync();
        if (sel) {
            const u32 r0 = lds[WO_RES + 4 * rank], r1 = lds[WO_RES + 4 * rank + 1], kind = lds[WO_RES + 4 * rank + 2];
            q.c0[0] = r0;
            q.c1[0] = (kind & QF_TWO) ? r1 : 0u;
            q.info[0] |= (u32)QF_EXP | kind;
        }
        AVK_WT_MARK(1)
    }
    if (nbest == 0) return AVK_ST_NO_RESULTS; /* :331 */
    out.n_opt = nbest;
    if (a.mode == 1) { /* merge_solver.rs:137-143: all_opt_haps[0].is_exact_match(); every tied optimum has the same total cost */
        out.ed1 = best == 0 ? 1u : 0u;
        return AVK_ST_OK;
    }

    /* ---- phase B for every tied optimum (waffle_solver.rs:169-261), as many searches at once as the working area holds queues; the first optimum with
     * the fewest flips wins (:264-265) */
    wv_sync();
    u32 *const opth = dyn; /* the optima's haplotype states leave the pool (through registers: source and destination overlap) */
    {
        u32 keep[8];
#pragma unroll
        for (u32 i = 0; i < 8; ++i) {
            const u32 k = lane + 64u * i;
            keep[i] = k < nbest * 8u ? nodes[(lds[WO_OPT + (k >> 3)] & 0xFFu) * W_NODE_W + (k & 7u)] : 0u;
        }
        wv_sync();
#pragma unroll
        for (u32 i = 0; i < 8; ++i) {
            const u32 k = lane + 64u * i;
            if (k < nbest * 8u) opth[k] = keep[i];
        }
    }
    wv_sync();
    u32 *const qrows = dyn + nbest * 8u;
    const u32 q_room = (dyn_words - nbest * 8u) / (2u * (u32)W_QB_W);
    const u32 chunk = q_room < 32u ? q_room : 32u; /* optima per pass */
    if (chunk == 0) return AVK_WDEFER(13);
    u32 win = 0, best_total = 0xFFFFFFFFu, o_t0 = 0, o_q0 = 0, o_t1 = 0, o_q1 = 0;
    for (u32 kb = 0; kb < nbest && best_total != 0; kb += chunk) { /* the reference goes through the optima in order and stops at the first one without flips */
        const u32 nk = nbest - kb < chunk ? nbest - kb : chunk;
        int eB = 0;
        u32 rt = 0, rq = 0;
        if (lane < 2 * nk) {
            Hap h;
            hap_unpack(h, opth + 8 * kb + 4 * lane);
            if (h.ed == 0 && h.nskip == 0) { /* truth == query with every ALT incorporated: the zero-flip path wins (exact_gt_optimizer.rs:169-192) */
                rt = h.t_alt;
                rq = h.q_alt;
            } else {
                WCtx cb = c;
                eB = phaseB(cb, qrows + lane * (u32)W_QB_W, h.t_alt, h.q_alt, rt, rq);
            }
        }
        for (u32 k = 0; k < nk; ++k) {
            const int e0 = (int)wv_readlane((u32)eB, 2 * k), e1 = (int)wv_readlane((u32)eB, 2 * k + 1);
            if (e0 == WD_DEFER) return AVK_WDEFER(13);
            if (e0 < 0) return -e0 - 100;
            if (e1 == WD_DEFER) return AVK_WDEFER(14);
            if (e1 < 0) return -e1 - 100;
            const u32 total = (u32)e0 + (u32)e1;
            if (total < best_total) {
                best_total = total;
                win = kb + k;
                o_t0 = wv_readlane(rt, 2 * k), o_q0 = wv_readlane(rq, 2 * k), o_t1 = wv_readlane(rt, 2 * k + 1), o_q1 = wv_readlane(rq, 2 * k + 1);
                if (total == 0) break;
            }
        }
        wv_sync();
    }
    AVK_WT_MARK(3)
    Hap wn[2];
    hap_unpack(wn[0], opth + 8 * win);
    hap_unpack(wn[1], opth + 8 * win + 4);
    out.ed1 = wn[0].ed;
    out.ed2 = wn[1].ed;

    /* ---- phase C: compare_expected_observed (:296-327) + per-call outputs, lane s = call slot s */
    u32 ex = 0, ob = 0, l_bad = 0;
    {
        const u32 side = (lane >> 3) & 1u, j = lane & 7u;
        const bool on = lane < (u32)WNS && j < (side ? Q : T);
        if (on) {
            const u32 b0 = ((side ? wn[0].q_alt : wn[0].t_alt) >> j) & 1u, b1 = ((side ? wn[1].q_alt : wn[1].t_alt) >> j) & 1u;
            const u32 o0 = ((side ? o_q0 : o_t0) >> j) & 1u, o1 = ((side ? o_q1 : o_t1) >> j) & 1u;
            ex = b0 + b1, ob = o0 + o1;
            if (ex == 0) l_bad = AVK_ST_VARIANT_METRICS;
            else if (ex < ob) l_bad = AVK_ST_TRUTH_FP;
            u32 cls = ex == ob ? AVK_CLASS_TP : AVK_CLASS_FN;
            u32 ea = ex, oa = ob;
            if (side) {
                if (cls == AVK_CLASS_FN) cls = AVK_CLASS_FP;
                ea = ob;
                oa = ex;
            }
            const u32 rz = b0 && b1 ? AVK_ZYG_HOM_ALT : (b0 ? AVK_ZYG_PHASED_HET10 : AVK_ZYG_PHASED_HET01);
            a.var_out[v_off + (side ? T + j : j)] = ea | (oa << 8) | (cls << 16) | (rz << 24);
        }
    }
    const u32 bad = wv_max_u32(l_bad);
    if (bad) return (int)bad;
    const u32 ex_lo = (u32)wv_ballot(ex & 1u), ex_hi = (u32)wv_ballot(ex & 2u), ob_lo = (u32)wv_ballot(ob & 1u), ob_hi = (u32)wv_ballot(ob & 2u);

    /* add_basepair_stats (:335-449): per haplotype X = 2 ed(ref, truth), Y = 2 ed(ref, query), Z = 2 ed(truth, query); lane (haplotype, side).
     * Every lane that aligns gets a scratch of its own in the working area, by its rank among the lanes that do. */
    const u32 SUPMASK = (1u << AVK_VT_SNV) | (1u << AVK_VT_INSERTION) | (1u << AVK_VT_DELETION) | (1u << AVK_VT_INDEL) | (1u << AVK_VT_TR_CONTRACTION) |
                        (1u << AVK_VT_TR_EXPANSION) | (1u << AVK_VT_SV_DELETION) | (1u << AVK_VT_SV_INSERTION);
    out.present = types | SUPMASK;
    wv_sync();
    const bool coop = ed_bound > (u32)W_COOP_MIN;
    const u32 scr_stride = scr_w | 1u, scr_slots = dyn_words / scr_stride;
    if (!coop && scr_slots < 4) return AVK_WDEFER(15);
    WCtx cs = c;
    cs.coop = coop ? 1u : 0u;
    cs.wfp = dyn + (lane & 3u) * scr_stride;
    int e_ref = 0;
    if (lane < 4) {
        const Hap &h = (lane >> 1) ? wn[1] : wn[0];
        const u32 side = lane & 1u, alt = side ? h.q_alt : h.t_alt;
        if (alt) e_ref = lane::ed_to_ref(cs, side, alt, side ? h.q_len : h.t_len);
    }
    if (coop && !resolve_alignments(cs, e_ref)) return AVK_WDEFER(15);
    if (wv_ballot(e_ref < 0) != 0) return AVK_WDEFER(15);
    const u32 X0 = 2u * wv_readlane((u32)e_ref, 0), Y0 = 2u * wv_readlane((u32)e_ref, 1), X1 = 2u * wv_readlane((u32)e_ref, 2), Y1 = 2u * wv_readlane((u32)e_ref, 3);
    const u32 tp0 = (X0 + Y0 - 2u * wn[0].ed) / 2u, tp1 = (X1 + Y1 - 2u * wn[1].ed) / 2u;
    /* Alignments of the per-type groups (:383-445), before anything is added to the tally (they can still hand the region over): lane (type, side,
     * haplotype).  A side that has calls of the type AND calls of other types is compared with only the type's calls applied:
     * x = ed(ref, filtered side), z = ed(filtered side, other side as it is). */
    {
        const u32 vt = lane >> 2, side = (lane >> 1) & 1u, hh = lane & 1u;
        int x = 0, z = 0;
        bool store = false;
        u32 mask_g = 0;
        const u32 cnt = side ? Q : T;
        if (vt < (u32)AVK_N_VARIANT_TYPES && ((types & SUPMASK) >> vt) & 1u) {
            for (u32 j = 0; j < cnt; ++j)
                if (((lds[WO_VW0 + WMV * side + j] >> 24) & 0xFu) == vt) mask_g |= 1u << j;
            store = mask_g != 0 && mask_g != (1u << cnt) - 1u; /* else: none of the type, or nothing but the type: no filtering */
        }
        const u64 sm = wv_ballot(store);
        if (!coop && (u32)avk_popc64(sm) > scr_slots) return AVK_WDEFER(16);
        wv_sync();
        cs.wfp = dyn + (coop ? 0u : (u32)avk_popc64(sm & below) * scr_stride);
        const Hap &h = hh ? wn[1] : wn[0];
        const u32 st = c.seq_id(0, h.t_alt), sq = c.seq_id(1, h.q_alt);
        const u32 Xh = hh ? X1 : X0, Yh = hh ? Y1 : Y0;
        const u32 alt = side ? h.q_alt : h.t_alt, m = alt & mask_g;
        const bool part = store && m && m != alt; /* some of the side's calls are filtered away on this haplotype: two alignments */
        const u32 sf = part ? c.seq_id(side, m) : 0u, fl = part ? c.seq_len(sf) : 0u;
        if (store) {
            x = 0, z = (int)((side ? Xh : Yh) / 2); /* nothing left of the side: it is the reference window */
            if (m && m == alt) {                    /* nothing filtered away on this haplotype: the side as it is */
                x = (int)((side ? Yh : Xh) / 2);
                z = (int)h.ed;
            }
        }
        if (part) x = lane::ed_to_ref(cs, side, m, fl);
        if (coop && !resolve_alignments(cs, x)) return AVK_WDEFER(16);
        if (part) {
            /* equal haplotypes with everything applied, one call filtered away: the other side's string is the filtered string plus that call */
            const u32 gone = alt ^ m;
            z = -1;
            if (h.ed == 0 && h.nskip == 0 && (gone & (gone - 1)) == 0 && c.seq_fail(side ? sq : st) == 0 && c.seq_fail(sf) == 0)
                z = lane::one_call_distance(cs, WMV * side + (u32)__builtin_ctz(gone));
            if (z < 0) z = side ? wfa_ed(cs, st, h.t_len, sf, fl) : wfa_ed(cs, sf, fl, sq, h.q_len);
        }
        if (coop && !resolve_alignments(cs, z)) return AVK_WDEFER(16);
        if (wv_ballot(store && (x < 0 || z < 0)) != 0) return AVK_WDEFER(16);
        if (store) lds[WO_FILT + lane] = (u32)x | ((u32)z << 16);
    }
    wv_sync();

    AVK_WT_MARK(4)
    /* ---- the metric groups, lane g = group g: the joint one and one per call type of the region */
    const u32 gmask = 1u | (types << 1);
    const bool g_on = lane < (u32)AVK_N_GROUPS && ((gmask >> lane) & 1u);
    lane::Group22 G;
#pragma unroll
    for (int i = 0; i < AVK_N_FIELDS; ++i) G.f[i] = 0;
    u32 l_err = 0;
    if (g_on) {
        const u32 g = lane;
        u32 tot_t = 0, tot_q = 0, tcount = 0, qcount = 0, tmask_g = 0, qmask_g = 0;
        for (u32 s = 0; s < (u32)WNS; ++s) {
            const bool on = (s & 7u) < (s < (u32)WMV ? T : Q);
            const u32 w0 = lds[WO_VW0 + s], w1 = lds[WO_VW1 + s];
            const u32 vt = (w0 >> 24) & 0xFu, zy = (w0 >> 28) & 7u;
            if (!on || (g != 0 && vt != g - 1)) continue;
            const u32 exs = ((ex_lo >> s) & 1u) | (((ex_hi >> s) & 1u) << 1), obs = ((ob_lo >> s) & 1u) | (((ob_hi >> s) & 1u) << 1);
            if (s >= (u32)WMV) lane::g_add<true>(G, w1 & 0xFFu, exs, obs);
            else lane::g_add<false>(G, w1 & 0xFFu, exs, obs);
            const u32 cntz = zy == AVK_ZYG_HOM_ALT ? 2u : ((zy == AVK_ZYG_UNPHASED_HET || zy == AVK_ZYG_PHASED_HET01 || zy == AVK_ZYG_PHASED_HET10) ? 1u : 0u);
            const u32 val = cntz * ((w1 >> 8) & 0xFFFFu);
            if (s < (u32)WMV) {
                tot_t += val;
                tcount += 1;
                tmask_g |= 1u << s;
            } else {
                tot_q += val;
                qcount += 1;
                qmask_g |= 1u << (s - WMV);
            }
        }
        if (g == 0) {
            G.f[AVK_F_BP_TRUTH_TP] += tp0 + tp1;
            G.f[AVK_F_BP_TRUTH_FN] += X0 - tp0 + 2 * wn[0].t_skip + X1 - tp1 + 2 * wn[1].t_skip; /* + skip metrics :378-381 */
            G.f[AVK_F_BP_QUERY_TP] += tp0 + tp1;
            G.f[AVK_F_BP_QUERY_FP] += Y0 - tp0 + 2 * wn[0].q_skip + Y1 - tp1 + 2 * wn[1].q_skip;
        } else if ((SUPMASK >> (g - 1)) & 1u) { /* :383-445: one side filtered to the type against the other side as it is */
            for (u32 hh = 0; hh < 2; ++hh) {
                const Hap &h = hh ? wn[1] : wn[0];
                const u32 Xh = hh ? X1 : X0, Yh = hh ? Y1 : Y0, tph = hh ? tp1 : tp0;
                u32 q_tp = 0, q_fp = 0, t_tp = 0, t_fn = 0;
                if (qcount) {
                    if (qcount == Q) {
                        q_tp = tph;
                        q_fp = Yh - tph + 2 * h.q_skip;
                    } else { /* some of the side's calls: the alignments were made above */
                        const u32 sf = c.seq_id(1, h.q_alt & qmask_g);
                        const u32 e = lds[WO_FILT + 4 * (g - 1) + 2 + hh];
                        const u32 y2 = e & 0xFFFFu, z2 = e >> 16;
                        const u32 tp2 = (Xh + 2u * y2 - 2u * z2) / 2u;
                        q_tp = tp2;
                        q_fp = 2u * y2 - tp2 + 2 * c.seq_fail(sf);
                    }
                }
                if (tcount) {
                    if (tcount == T) {
                        t_tp = tph;
                        t_fn = Xh - tph + 2 * h.t_skip;
                    } else {
                        const u32 sf = c.seq_id(0, h.t_alt & tmask_g);
                        const u32 e = lds[WO_FILT + 4 * (g - 1) + hh];
                        const u32 x2 = e & 0xFFFFu, z2 = e >> 16;
                        const u32 tp2 = (2u * x2 + Yh - 2u * z2) / 2u;
                        t_tp = tp2;
                        t_fn = 2u * x2 - tp2 + 2 * c.seq_fail(sf);
                    }
                }
                G.f[AVK_F_BP_TRUTH_TP] += t_tp;
                G.f[AVK_F_BP_TRUTH_FN] += t_fn;
                G.f[AVK_F_BP_QUERY_TP] += q_tp;
                G.f[AVK_F_BP_QUERY_FP] += q_fp;
            }
        }
        { /* add_record_basepair_stats (:455-522) */
            const u32 tfn = G.f[AVK_F_BP_TRUTH_FN], qfp = G.f[AVK_F_BP_QUERY_FP];
            const u32 ttp = 2 * tot_t - tfn, qtp = 2 * tot_q - qfp;
            if (g == 0 && (ttp < G.f[AVK_F_BP_TRUTH_TP] || qtp < G.f[AVK_F_BP_QUERY_TP])) l_err = AVK_ST_RECORD_BP;
            G.f[AVK_F_RBP_TRUTH_TP] += ttp;
            G.f[AVK_F_RBP_TRUTH_FN] += tfn;
            G.f[AVK_F_RBP_QUERY_TP] += qtp;
            G.f[AVK_F_RBP_QUERY_FP] += qfp;
        }
    }
    const u32 rerr = wv_max_u32(l_err);
    if (rerr) return (int)rerr;
    /* SummaryWriter::add_comparison_benchmark (writers/summary.rs:146-163): the region's nonzero counters (a handful of the 286) go to a partial tally, lane g
     * its group's; the optional per-region outputs */
    u32 *gm_out = a.group_metrics ? a.group_metrics + (u64)orig * AVK_N_GROUPS * AVK_N_FIELDS : (u32 *)0;
    if (gm_out) {
        for (u32 i = lane; i < (u32)(AVK_N_GROUPS * AVK_N_FIELDS); i += 64) gm_out[i] = 0;
        wv_sync();
    }
    if (g_on) {
        const u32 g = lane;
#pragma unroll
        for (int i = 0; i < AVK_N_FIELDS; ++i) {
            const u32 v = G.f[i];
            if (!v) continue;
            avk_atomic_add_u64_global(part + g * AVK_N_FIELDS + i, v);
            if (gm_out) gm_out[g * AVK_N_FIELDS + i] = v;
        }
        if (a.bp_out) { /* compact BASEPAIR groups: the joint group, then the call types of the region in type order */
            avk_u4 w;
            w.x = G.f[AVK_F_BP_TRUTH_TP], w.y = G.f[AVK_F_BP_TRUTH_FN], w.z = G.f[AVK_F_BP_QUERY_TP], w.w = G.f[AVK_F_BP_QUERY_FP];
            *(avk_u4 *)(a.bp_out + 4 * ((u64)a.bp_off[orig] + (u32)__builtin_popcount(gmask & ((1u << g) - 1u)))) = w;
        }
    }
    AVK_WT_MARK(5)
    return AVK_ST_OK;
}

/* One persistent wave: claims regions of the launch's list one at a time (AvkKernelArgs: work_list / n_work_dev or records work_base .. work_base +
 * n_work - 1, work_counter), appends what it cannot take to overflow_list.  lds = the wave's LDS (WideArgs::lds_words words). */
template <bool LAZY> AVK_DEV void wide_worker(const AvkKernelArgs &a, const WideArgs &wa, u32 wave_id, u32 *lds) {
    const u32 lane = (u32)wv_lane();
    const u32 n_work = a.n_work_dev ? wv_uni(*a.n_work_dev) : a.n_work;
    u32 n_ok = 0, n_err = 0;
    u64 *part = a.tally + (u64)(wave_id % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE;
#ifdef AVK_WIDE_TIMING
    WTime wt;
    for (int k = 0; k < 16; ++k) wt.t[k] = 0;
    const u64 t_wave0 = avk_clock();
    u64 t_lazy = 0, t_max = 0;
    (void)t_wave0, (void)t_lazy, (void)t_max;
#endif
    for (;;) {
        u32 idx = 0xFFFFFFFFu;
        if (lane == 0) {
            const u32 seen = avk_ld_agent_u32(a.work_counter);
            if (seen < n_work) idx = avk_atomic_add_u32_global(a.work_counter, 1u);
        }
        idx = wv_uni(wv_shfl(idx, 0));
        if (idx == 0xFFFFFFFFu || idx >= n_work) break;
        const u32 r = a.work_list ? wv_uni(a.work_list[idx]) : a.work_base + idx;
        if (LAZY && a.lazy_dp && r >= a.lazy_from) { /* a lane-class region of a device-packed batch: its record and blob are written now, by this wave */
#ifdef AVK_WIDE_TIMING
            const u64 t_l0 = avk_clock();
#endif
            if (lane == 0) dp::dp_region_record(*(const dp::DpArgs *)a.lazy_dp, r);
            wv_sync();
#ifdef AVK_WIDE_TIMING
            t_lazy += avk_clock() - t_l0;
#endif
        }
        lane::LaneOut out;
        out.ed1 = out.ed2 = out.n_opt = out.present = 0;
#ifdef AVK_WIDE_TIMING
        const u64 t_region0 = avk_clock(), r8 = wt.t[8], r9 = wt.t[9];
        wt.last = t_region0;
#endif
        const int st = solve_wide(a, wa, r, lds, part, out AVK_WT_PASS);
        wv_sync();
#ifdef AVK_WIDE_TIMING
        wt.t[6] += avk_clock() - t_region0;
        { /* the longest region of the wave: ticks / 16 << 32 | rounds << 20 | pops << 8 | calls */
            const u64 dt = (avk_clock() - t_region0) >> 4;
            const u64 rec = (dt << 32) | (((wt.t[8] - r8) & 0xFFFu) << 20) | (((wt.t[9] - r9) & 0xFFFu) << 8) | ((a.regions[r].t_cnt + a.regions[r].q_cnt) & 0xFFu);
            if (rec > t_max) t_max = rec;
        }
        wt.t[7] += 1;
        wt.t[11] += st == WD_DEFER ? 1u : 0u;
#endif
        if (st == WD_SKIP) continue;
        if (st == WD_DEFER) {
            if (lane == 0) {
                const u32 slot_o = avk_atomic_add_u32_global(a.overflow_count, 1u);
                a.overflow_list[slot_o] = r;
            }
            continue;
        }
        if (st != AVK_ST_OK) {
            write_failed_region(a, r, st);
            n_err += 1;
            continue;
        }
        write_region_record(a, wv_uni(a.regions[r].orig), 0, out.ed1, out.ed2, out.n_opt, out.present);
        n_ok += 1;
    }
    if (lane == 0) {
        if (n_ok) avk_atomic_add_u64_global(part + AVK_TALLY_SOLVED, n_ok);
        if (n_err) avk_atomic_add_u64_global(part + AVK_TALLY_ERRORS, n_err);
        if (n_ok + n_err) avk_atomic_add_u64_global(part + AVK_TALLY_WIDE_SOLVED, n_ok + n_err);
#ifdef AVK_WIDE_TIMING
#ifdef AVK_WIDE_TRACE /* this build reports the hand-back launch alone; [12] the waves that took a region, [13] their lifetimes, [14] the time spent writing records on demand, [15] the longest region */
        if (LAZY && wt.t[7]) {
            wt.t[12] = 1, wt.t[13] = avk_clock() - t_wave0, wt.t[14] = t_lazy;
            for (int k = 0; k < 15; ++k) avk_atomic_add_u64_global(part + AVK_TALLY_LEN + 5 + k, wt.t[k]);
            avk_atomic_max_u64_global(a.tally + AVK_TALLY_LEN + 5 + 15, t_max); /* (one copy: the reduce sums the copies) */
        }
#else
        for (int k = 0; k < 16; ++k) avk_atomic_add_u64_global(part + AVK_TALLY_LEN + 5 + k, wt.t[k]);
#endif
#endif
    }
}

} // namespace wide
} // namespace avk
#endif
