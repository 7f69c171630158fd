/*
 * avk_solver.inl — the per-region compare solver as wave-cooperative device code.
 *
 * One 64-lane wavefront solves one region end to end (reference solve_compare_region,
 * src/waffle_solver.rs:122-284):
 *   phase A  phasing search        optimize_sequences   src/query_optimizer.rs:166-365
 *   phase B  genotype assignment   optimize_gt_alleles  src/exact_gt_optimizer.rs:108-357 (x2 per tied optimum)
 *   phase C  metrics               compare_expected_observed / add_basepair_stats /
 *                                  add_record_basepair_stats  src/waffle_solver.rs:296-522
 * on top of the dynamic wavefront aligner (src/dwfa/dynamic_wfa.rs:23-276) and the haplotype
 * builders (src/dwfa/haplotype_dwfa.rs:17-245).
 *
 * MI355X mapping (not a translation of the reference's per-node heap objects):
 *   - control flow is wave-uniform; all search bookkeeping (queue length, ids, quotas, best cost)
 *     lives in scalar registers;
 *   - the wavefront diagonals are extended by lane groups: 64/G diagonals at a time, G lanes
 *     comparing G consecutive bases of one diagonal per step, one ballot + ctz per step;
 *   - the best-first queues are flat (key, slot) arrays scanned 64 entries at a time and reduced
 *     with a cross-lane min — exact pop order without a heap; keys are unique so order is total;
 *   - nodes are fixed-stride records in a per-wave workspace: the LDS slice of the wave for the
 *     common small regions, the wave's private HBM slice for larger ones, a big HBM slice in a
 *     second pass for the long tail.  A tier that runs out of room reports overflow and the
 *     region restarts on the next tier, so results never depend on the tier;
 *   - haplotype sequences, wavefronts and allele bit-sets are copied lane-parallel as words.
 *
 * Every cross-lane exchange through memory is separated by wv_sync(); values that steer
 * control flow are made uniform with wv_uni().
 */
#ifndef AVK_SOLVER_INL
#define AVK_SOLVER_INL

#include "avk_dev_types.h"
#include "avk_wave.h"
#include "avk_devpack.inl" /* dp_region_record: region records of device-packed batches are written on demand */

namespace avk {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

enum { AL_REF = 1, AL_ALT = 2 };
enum { RS_OVERFLOW = -1 }; /* internal: this tier's workspace is too small, retry on the next */
enum { RS_CUT = -2 };      /* internal: phase B stopped because its answer can no longer matter */
#ifndef AVK_STATIC_PCT
#define AVK_STATIC_PCT 75
#endif
#ifndef AVK_CLAIM
#define AVK_CLAIM 2u /* regions claimed per work-counter atomic */
#endif

#define AVK_ALIGN8(x) (((x) + 7u) & ~(u64)7u)
#define AVK_ALIGN16(x) (((x) + 15u) & ~(u64)15u)
/* a node = NODE_HDR bytes (id, auxiliary word) + its haplotype record(s); records start on 16-byte boundaries so their
 * 12 header words move as three 16-byte LDS accesses */
#define NODE_HDR 16u

/* optional per-phase cycle accounting (profiling builds: -DAVK_PHASE_TIMING) */
#ifdef AVK_PHASE_TIMING
#define AVK_T_DECL u64 t_phase_ = avk_clock();
#define AVK_T_MARK(c, k)                    \
    {                                       \
        const u64 n_ = avk_clock();         \
        (c).tphase[k] += n_ - t_phase_;     \
        t_phase_ = n_;                      \
    }
#elif defined(AVK_STOP_AFTER)
/* ablation builds: the region ends (as Ok, with whatever has been computed) after phase AVK_STOP_AFTER;
 * timing experiments only, results are meaningless */
#define AVK_T_DECL
#define AVK_T_MARK(c, k) \
    if ((k) == AVK_STOP_AFTER) return AVK_ST_OK;
#else
#define AVK_T_DECL
#define AVK_T_MARK(c, k)
#endif
#ifndef AVK_PHASE_TIMING
#define AVK_TA_DECL
#define AVK_TA_MARK(c, k)
#else
#define AVK_TA_DECL AVK_T_DECL
#define AVK_TA_MARK(c, k) AVK_T_MARK(c, k)
#endif

/* hap record: 10 header words, then alt bit-sets, wavefront, two sequences */
enum { H_T_REFPOS = 0, H_Q_REFPOS, H_T_LEN, H_Q_LEN, H_T_SKIP, H_Q_SKIP, H_ED, H_T_NAL, H_Q_NAL, H_NSKIP, H_D0, H_PAD, H_WORDS };

struct HapHdr {
    u32 t_refpos, q_refpos, t_len, q_len, t_skip, q_skip, ed, t_nal, q_nal;
    u32 nskip; /* ALT alleles that could not be incorporated (either side); the skip DISTANCE can be 0 for REF == ALT */
    u32 d0;    /* while ed == 0 the whole wavefront is this one offset and lives here, not in wf[] */
};

/* local variant record in the workspace */
struct LVar { /* 7 words; type_zyg = type | zyg << 8 */
    u32 rel_pos, a0_len, a1_len, a_off, raw_space, alt_ed, type_zyg;
};

/* a variant record fetched for the whole wave: every field is made scalar, so the arithmetic and the
 * branches that depend on it stay on the scalar unit */
struct UVar {
    u32 rel_pos, a0_len, a1_len, a_off, raw_space, alt_ed, type, zyg;
};

struct TeamBox;
struct Ctx {
    /* region */
    u32 L, T, Q, N;
    TeamBox *team;          /* the workgroup's team (avk_region_kernel_team) or NULL: this wave alone */
    u32 team_gen;           /* generation of the last batch of jobs posted */
    u32 team_waves;         /* waves of the team that take jobs (1: the owner alone) */
    u32 *team_dbg;          /* seven words of device counters the team writes its progress to (tools/gpu_team_probe.py reads them while a launch is in flight) */
    u32 team_mode;          /* AvkKernelArgs::team */
    u64 team_scratch_bytes; /* bytes of scratch every sibling has for the alignments of the metrics (its own workspace slice) */
    u8 *ws;        /* base of this wave's workspace */
    const u8 *ref; /* window bytes (workspace copy) */
    LVar *vars;    /* [N]: truth 0..T-1, query T..N-1 */
    u8 *alle;
    const AvkOrdVar *ovars; /* [N] the variants in processing order (query_optimizer.rs:372-381), each with its sync point */
    u32 *counts;   /* [8] truth | query << 16 variants of each type of AVK_SUP_TYPES */
    u32 *bucket;   /* [N+1] */
    /* capacities */
    u32 seqcap, wfcap, alw;
    u32 hapA_bytes, nodeA_bytes, hapB_bytes, nodeB_bytes;
    /* node pool + queue */
    u8 *pool;
    u64 pool_bytes;
    u32 node_bytes; /* stride of the phase that is running */
    u8 *pool_base;  /* first node of the running phase */
    u32 pool_cap, pool_used, nfree;
    u32 *freelist;
    u64 *qkeys;
    u32 *qslots;
    u32 qn, qcap;
    /* register-resident form of the queue / free list / per-depth counters, used while the phase's pool has at
     * most 64 nodes (and at most 64 depths): lane j holds queue entry j, lane d the counter of depth d, the free
     * slots are a scalar bit mask — no LDS traffic for the search bookkeeping */
    bool regq, regb;
    u64 rq_key;
    u32 rq_slot, rbucket;
    u64 freemask;
    /* scratch */
    u32 *wfs;
    u32 wfs_cap;
    u8 *seq_a;
    u64 cscratch_off, pool_off; /* byte offsets of the metrics scratch and of the pool in the workspace */
    u32 *gm, *gq; /* [13*22] metrics blocks */
    u32 *optlist; /* tied optima (node indices) */
    u32 optcap;
    u64 *bres;    /* phase-B results: [2 cand][2 hap][2 side][alw] alt bit-sets */
    u64 *memo;    /* phase-B memo: [memo_cap][4*alw] input and result bit-sets */
    u32 *memo_err;
    u32 memo_cap;
    u32 max_branch;
    u32 best_cost; /* total cost shared by the tied optima of phase A */
#ifdef AVK_PHASE_TIMING
    u64 tphase[16];
#endif
};

/* ------------------------------------------------------------------------------------------ */
/* small lane-parallel helpers                                                                */
/* ------------------------------------------------------------------------------------------ */
AVK_DEV void copy_bytes(u8 *dst, const u8 *src, u32 n) {
    for (u32 i = (u32)wv_lane(); i < n; i += 64) dst[i] = src[i];
}
AVK_DEV void copy_words(u32 *dst, const u32 *src, u32 n) {
    for (u32 i = (u32)wv_lane(); i < n; i += 64) dst[i] = src[i];
}
/* the same for long runs (the sequences of a search node in a window of kilobases: tens of KB per node copy, in HBM): 16 bytes per lane and access, four accesses
 * of a lane in flight — a wave moves 4 KB per memory round trip instead of 256 bytes.  dst and src sit at the same offset of two records, so they are aligned alike. */
AVK_DEV void copy_words_long(u32 *dst, const u32 *src, u32 n) {
    if (n < 512u || ((((uintptr_t)dst) ^ ((uintptr_t)src)) & 15u)) {
        copy_words(dst, src, n);
        return;
    }
    const u32 lane = (u32)wv_lane();
    const u32 head = (u32)((16u - ((uintptr_t)src & 15u)) & 15u) >> 2;
    if (lane < head) dst[lane] = src[lane];
    const avk_u4 *s4 = (const avk_u4 *)(src + head);
    avk_u4 *d4 = (avk_u4 *)(dst + head);
    const u32 n4 = (n - head) >> 2;
    u32 i = lane;
    for (; i + 192u < n4; i += 256u) {
        const avk_u4 v0 = s4[i], v1 = s4[i + 64u], v2 = s4[i + 128u], v3 = s4[i + 192u];
        d4[i] = v0, d4[i + 64u] = v1, d4[i + 128u] = v2, d4[i + 192u] = v3;
    }
    for (; i < n4; i += 64u) d4[i] = s4[i];
    const u32 done = head + 4u * n4;
    if (done + lane < n) dst[done + lane] = src[done + lane];
}
AVK_DEV void zero_words(u32 *dst, u32 n) {
    for (u32 i = (u32)wv_lane(); i < n; i += 64) dst[i] = 0;
}
AVK_DEV void st32(u32 *p, u32 v) {
    if (wv_lane() == 0) *p = v;
}
AVK_DEV void st64(u64 *p, u64 v) {
    if (wv_lane() == 0) *p = v;
}
AVK_DEV u32 ld32u(const u32 *p) { return wv_uni(*p); }
/* loads n consecutive words first and makes them scalar afterwards: the loads go out back to back and
 * complete under one wait instead of one LDS round trip per field */
template <int N> AVK_DEV void ldvec_u(const u32 *p, u32 (&out)[N]) {
    u32 t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = p[k];
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = wv_uni(t[k]);
}
/* one AvkOrdVar: two 16-byte LDS reads under one wait; also returns the step's sync point and the variant's index */
AVK_DEV UVar load_ovar(const AvkOrdVar *ov, u32 depth, u32 &sync, u32 &vi) {
    const avk_u4 *q = (const avk_u4 *)(ov + depth);
    const avk_u4 a = q[0], b = q[1];
    UVar v;
    v.rel_pos = wv_uni(a.x);
    v.a0_len = wv_uni(a.y);
    v.a1_len = wv_uni(a.z);
    v.a_off = wv_uni(a.w);
    v.raw_space = 0;
    v.alt_ed = wv_uni(b.x);
    const u32 tz = wv_uni(b.y);
    v.type = tz & 0xFF;
    v.zyg = (tz >> 8) & 0xFF;
    sync = wv_uni(b.z);
    vi = wv_uni(b.w);
    return v;
}
AVK_DEV UVar load_var(const LVar *vars, u32 i) {
    u32 w[7];
    ldvec_u<7>((const u32 *)(vars + i), w);
    UVar v;
    v.rel_pos = w[0];
    v.a0_len = w[1];
    v.a1_len = w[2];
    v.a_off = w[3];
    v.raw_space = w[4];
    v.alt_ed = w[5];
    v.type = w[6] & 0xFF;
    v.zyg = (w[6] >> 8) & 0xFF;
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* dynamic wavefront aligner — src/dwfa/dynamic_wfa.rs                                         */
/* wf[i] = symbols of O consumed on diagonal i, baseline offset = wf[i] + ed - i (:114)        */
/* ------------------------------------------------------------------------------------------ */

/* extend (:94-130): every diagonal slides while the bases agree and both offsets are in range.
 * 64/G diagonals are processed at once, G lanes per diagonal. */
AVK_DEV void dw_extend(u32 *wf, u32 ed, const u8 *B, u32 bl, const u8 *O, u32 ol) {
    const u32 nd = 2 * ed + 1;
    const u32 lane = (u32)wv_lane();
    u32 gs; /* log2(G) */
    if (nd == 1) gs = 6;
    else if (nd <= 4) gs = 4;
    else if (nd <= 8) gs = 3;
    else if (nd <= 16) gs = 2;
    else if (nd <= 32) gs = 1;
    else gs = 0;
    const u32 G = 1u << gs, ngrp = 64u >> gs;
    const u32 grp = lane >> gs, li = lane & (G - 1);
    const u64 gmask = gs == 6 ? ~0ull : ((1ull << G) - 1);
    if (gs <= 2) { /* wide fronts (more than 8 diagonals): few lanes per diagonal, so every lane compares EIGHT bases per step, its sixteen reads in flight together —
                      a diagonal of a window of kilobases slides hundreds of bases between two edits, a memory round trip per step */
        enum { K = 8 };
        for (u32 base = 0; base < nd; base += ngrp) {
            const u32 i = base + grp;
            const bool active = i < nd;
            u32 d = active ? wf[i] : 0;
            bool done = !active;
            for (;;) {
                u32 run = 0;
                if (!done) {
                    const u32 oo = d + li * K, bo = d + ed - i + li * K;
                    u8 xb[K], xo[K];
#pragma unroll
                    for (u32 k = 0; k < K; ++k) {
                        const bool in = bo + k < bl && oo + k < ol;
                        xb[k] = in ? B[bo + k] : (u8)0;
                        xo[k] = in ? O[oo + k] : (u8)1; /* past an end: a difference */
                    }
#pragma unroll
                    for (u32 k = 0; k < K; ++k) run += (run == k && xb[k] == xo[k]) ? 1u : 0u;
                }
                const u64 m = wv_ballot(done || run < K);
                const u64 gm = (m >> (grp * G)) & gmask;
                const u32 l0 = gm ? (u32)avk_ctz64(gm) : 0u;
                const u32 r0 = G == 1 ? run : wv_shfl(run, (grp << gs) + l0); /* the run of the group's first lane that stops */
                if (!done) {
                    d += gm ? l0 * K + r0 : G * K;
                    done = gm != 0;
                }
                if (wv_ballot(!done) == 0) break;
            }
            if (active && li == 0) wf[i] = d;
        }
        wv_sync();
        return;
    }
    for (u32 base = 0; base < nd; base += ngrp) {
        const u32 i = base + grp;
        const bool active = i < nd;
        u32 d = active ? wf[i] : 0;
        bool done = !active;
        for (;;) {
            bool match = false;
            if (!done) {
                const u32 oo = d + li, bo = d + ed - i + li;
                match = bo < bl && oo < ol && B[bo] == O[oo];
            }
            const u64 m = wv_ballot(!match);
            const u64 gm = (m >> (grp * G)) & gmask;
            if (!done) {
                d += gm ? (u32)avk_ctz64(gm) : G;
                done = gm != 0;
            }
            if (wv_ballot(!done) == 0) break;
        }
        if (active && li == 0) wf[i] = d;
    }
    wv_sync();
}

/* increase_edit_distance (:140-173) without the re-extend: new[k] = max(old[k], old[k-1]+1, old[k-2]+1),
 * no clipping to the sequence lengths.  In place, top chunk first. */
AVK_DEV void dw_bump(u32 *wf, u32 old_ed) {
    const u32 nd = 2 * old_ed + 1, nn = nd + 2;
    const u32 lane = (u32)wv_lane();
    for (int c = (int)(((nn - 1) >> 6) << 6); c >= 0; c -= 64) {
        const u32 k = (u32)c + lane;
        u32 v = 0;
        if (k < nn) {
            if (k < nd) v = wf[k];
            if (k >= 1 && k - 1 < nd) {
                const u32 t = wf[k - 1] + 1;
                v = t > v ? t : v;
            }
            if (k >= 2 && k - 2 < nd) {
                const u32 t = wf[k - 2] + 1;
                v = t > v ? t : v;
            }
        }
        wv_sync();
        if (k < nn) wf[k] = v;
        wv_sync();
    }
}

/* reached_baseline_end || reached_other_end (:220-231) as one ballot: some diagonal touches the end
 * of either sequence (the two maxima of :201-215 are only ever compared with the lengths) */
AVK_DEV bool dw_touches_end(const u32 *wf, u32 ed, u32 bl, u32 ol) {
    const u32 nd = 2 * ed + 1;
    bool any = false;
    for (u32 i = (u32)wv_lane(); i < nd; i += 64) {
        const u32 d = wf[i];
        any = any || d + ed - i >= bl || d >= ol;
    }
    return wv_ballot(any) != 0;
}

/* reached_full_diagonal (:237-245) */
AVK_DEV bool dw_full_diagonal(const u32 *wf, u32 ed, u32 bl, u32 ol) {
    const u32 nd = 2 * ed + 1;
    bool any = false;
    for (u32 i = (u32)wv_lane(); i < nd; i += 64) {
        const u32 d = wf[i];
        any = any || (d + ed - i >= bl && d >= ol);
    }
    return wv_ballot(any) != 0;
}

/* update (:68-84): extend, then raise the distance until EITHER end is touched.
 * returns 0 or RS_OVERFLOW when the wavefront no longer fits `cap` entries. */
AVK_DEV int dw_update(u32 *wf, u32 cap, u32 &ed, const u8 *B, u32 bl, const u8 *O, u32 ol) {
    dw_extend(wf, ed, B, bl, O, ol);
    while (!dw_touches_end(wf, ed, bl, ol)) {
        if (2 * ed + 3 > cap) return RS_OVERFLOW;
        dw_bump(wf, ed);
        ed += 1;
        dw_extend(wf, ed, B, bl, O, ol);
    }
    return 0;
}

/* finalize (:183-198) */
AVK_DEV int dw_finalize(u32 *wf, u32 cap, u32 &ed, const u8 *B, u32 bl, const u8 *O, u32 ol) {
    dw_extend(wf, ed, B, bl, O, ol);
    while (!dw_full_diagonal(wf, ed, bl, ol)) {
        if (2 * ed + 3 > cap) return RS_OVERFLOW;
        dw_bump(wf, ed);
        ed += 1;
        dw_extend(wf, ed, B, bl, O, ol);
    }
    return 0;
}

/* number of positions on which a[ia..] and b[ib..] agree before the first difference or either end */
AVK_DEV u32 seq_match_run(const u8 *a, u32 al, u32 ia, const u8 *b, u32 bl, u32 ib) {
    const u32 lane = (u32)wv_lane();
    const u32 ra = al > ia ? al - ia : 0, rb = bl > ib ? bl - ib : 0;
    const u32 lim = ra < rb ? ra : rb;
    u32 n = 0;
    while (n < lim) { /* four bases per lane and step, the eight reads in flight together: 256 positions per memory round trip */
        const u32 i0 = n + 4u * lane;
        u8 xa[4], xb[4];
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const bool in = i0 + k < lim;
            xa[k] = in ? a[ia + i0 + k] : (u8)0;
            xb[k] = in ? b[ib + i0 + k] : (u8)1; /* past the end: a difference */
        }
        u32 run = 0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) run += (run == k && xa[k] == xb[k]) ? 1u : 0u;
        const u64 m = wv_ballot(run < 4u);
        if (m) {
            const u32 l0 = (u32)avk_ctz64(m);
            n += 4u * l0 + wv_shfl(run, (int)l0);
            break;
        }
        n += 256;
    }
    return n;
}

/* wfa_ed (src/util/sequence_alignment.rs:9-13): a fresh DWFALite finalised on two complete strings, i.e. their
 * unit-cost edit distance (the reference asserts wfa_ed == edit_distance, :58-116; the oracle's wavefront code and
 * the full DP are fuzzed against each other).  Distances 0 and 1 — identical haplotypes, one substitution, one
 * single-base indel: nearly every call — are decided by one or two lane-parallel comparisons; anything else runs
 * the wavefront recurrence on the scratch front, starting from the common prefix.  < 0 = scratch overflow */
AVK_DEV int wfa_ed(const Ctx &c, const u8 *a, u32 al, const u8 *b, u32 bl) {
    wv_sync();
    const u32 lim = al < bl ? al : bl;
    const u32 d = seq_match_run(a, al, 0, b, bl, 0);
    if (d == lim) return (int)((al > bl ? al : bl) - lim); /* one is a prefix of the other */
    if (al == bl) {
        if (d + 1 + seq_match_run(a, al, d + 1, b, bl, d + 1) == al) return 1;
    } else if (al == bl + 1) {
        if (d + seq_match_run(a, al, d + 1, b, bl, d) == bl) return 1;
    } else if (bl == al + 1) {
        if (d + seq_match_run(a, al, d, b, bl, d + 1) == al) return 1;
    }
    st32(c.wfs, d);
    wv_sync();
    u32 ed = 0;
    if (dw_finalize(c.wfs, c.wfs_cap, ed, a, al, b, bl)) return RS_OVERFLOW;
    return (int)ed;
}

/* ------------------------------------------------------------------------------------------ */
/* hap records                                                                                */
/* ------------------------------------------------------------------------------------------ */
struct HapPtr {
    u32 *w;
    u64 *talt, *qalt;
    u32 *wf;
    u8 *tseq, *qseq;
};
AVK_DEV HapPtr hap_ptr(u8 *base, u32 alw, u32 wfcap, u32 seqcap) {
    HapPtr p;
    p.w = (u32 *)base;
    p.talt = (u64 *)(base + H_WORDS * 4);
    p.qalt = p.talt + alw;
    p.wf = (u32 *)(p.qalt + alw);
    p.tseq = (u8 *)(p.wf + wfcap);
    p.qseq = p.tseq + seqcap;
    return p;
}
/* the 12 header words of a haplotype record sit on a 16-byte boundary (NODE_HDR, AVK_ALIGN16 record sizes): three
 * 16-byte accesses instead of twelve 4-byte ones */
AVK_DEV HapHdr hap_load(const u32 *w) {
    const avk_u4 *q = (const avk_u4 *)w;
    const avk_u4 a = q[0], b = q[1], d = q[2];
    HapHdr h;
    h.t_refpos = wv_uni(a.x);
    h.q_refpos = wv_uni(a.y);
    h.t_len = wv_uni(a.z);
    h.q_len = wv_uni(a.w);
    h.t_skip = wv_uni(b.x);
    h.q_skip = wv_uni(b.y);
    h.ed = wv_uni(b.z);
    h.t_nal = wv_uni(b.w);
    h.q_nal = wv_uni(d.x);
    h.nskip = wv_uni(d.y);
    h.d0 = wv_uni(d.z);
    return h;
}
AVK_DEV void hap_store(u32 *w, const HapHdr &h) {
    if (wv_lane() == 0) {
        avk_u4 *q = (avk_u4 *)w;
        avk_u4 a, b, d;
        a.x = h.t_refpos;
        a.y = h.q_refpos;
        a.z = h.t_len;
        a.w = h.q_len;
        b.x = h.t_skip;
        b.y = h.q_skip;
        b.z = h.ed;
        b.w = h.t_nal;
        d.x = h.q_nal;
        d.y = h.nskip;
        d.z = h.d0;
        d.w = 0;
        q[0] = a;
        q[1] = b;
        q[2] = d;
    }
}
AVK_DEV void hap_init(const HapPtr &p, u32 alw) {
    if (wv_lane() == 0) {
        for (int k = 0; k < H_WORDS; ++k) p.w[k] = 0;
        for (u32 k = 0; k < alw; ++k) {
            p.talt[k] = 0;
            p.qalt[k] = 0;
        }
        p.wf[0] = 0;
    }
}

/* One haplotype step = HaplotypeDWFA::extend_variant without the DWFA update (haplotype_dwfa.rs:46-62,
 * :175-227), as ONE fused copy:
 *   other side:  copy_reference(sync)                                   -> segment O
 *   this side:   copy_reference(variant start)                          -> segment 1
 *                ALT and compatible (ref_pos <= start): allele1         -> segment 2   (else skip penalty)
 *                copy_reference(sync)                                   -> segment 3
 * All lengths are scalar; each lane moves the bytes whose index it owns, picking the source by range.
 * Returns the `success` flag of HaplotypeTracker::extend_variant.  `has_var` false = only the two
 * copy_reference(upto) calls of finalize_dwfa (:84-88). */
/* IS_TRUTH is a template parameter: as a run-time flag every header field it selects costs scalar selects on the way in and out */
template <bool is_truth> AVK_DEV bool hap_extend_seq_t(const Ctx &c, const HapPtr &p, HapHdr &h, bool has_var, const UVar &v, u32 allele, u32 sync) {
    const u32 lane = (u32)wv_lane();
    u32 &tl = is_truth ? h.t_len : h.q_len, &ol = is_truth ? h.q_len : h.t_len;
    u32 &trp = is_truth ? h.t_refpos : h.q_refpos, &orp = is_truth ? h.q_refpos : h.t_refpos;
    u32 &tskip = is_truth ? h.t_skip : h.q_skip, &tnal = is_truth ? h.t_nal : h.q_nal;
    u8 *ts = is_truth ? p.tseq : p.qseq, *os = is_truth ? p.qseq : p.tseq;
    u64 *talt = is_truth ? p.talt : p.qalt;

    const u32 n_o = orp < sync ? sync - orp : 0; /* other side up to the sync point */
    const u32 s_o = orp;
    u32 n1 = 0, n2 = 0, s1 = trp, rp = trp;
    bool ok = true;
    if (has_var) {
        if (rp < v.rel_pos) {
            n1 = v.rel_pos - rp;
            rp = v.rel_pos;
        }
        if (allele == AL_ALT) {
            if (rp <= v.rel_pos) {
                n2 = v.a1_len;
                rp = v.rel_pos + v.a0_len;
            } else {
                tskip += v.alt_ed; /* edit_distance(allele0, allele1), :199 — equal to the wavefront distance */
                h.nskip += 1;
                ok = false;
            }
        }
        if (lane == 0) { /* alleles.push(allele), :204 */
            const u64 bit = 1ull << (tnal & 63);
            if (allele == AL_ALT) talt[tnal >> 6] |= bit;
            else talt[tnal >> 6] &= ~bit;
        }
        tnal += 1;
    }
    const u32 s3 = rp;
    const u32 n3 = rp < sync ? sync - rp : 0;
    if (rp < sync) rp = sync;
    /* (sources — the window and the allele bytes — are the region's, destinations the record's: a genotype search dealt to a sibling wave keeps its records in
     * that wave's own slice, which may be any distance from the region's workspace: pointers, not 32-bit offsets from one base) */
    const u8 *const ref_p = c.ref, *const alle_p = c.alle + v.a_off + v.a0_len;
    u8 *const ts_p = ts + tl, *const os_p = os + ol;
    const u32 total = n_o + n1 + n2 + n3;
    for (u32 j = lane; j < total; j += 64) {
        const bool other = j < n_o;
        const u32 k = j - n_o; /* index on this side (meaningless when `other`) */
        const u8 *sp = other ? ref_p + s_o + j : (k < n1 ? ref_p + s1 + k : (k < n1 + n2 ? alle_p + (k - n1) : ref_p + s3 + (k - n1 - n2)));
        u8 *dp = other ? os_p + j : ts_p + k;
        *dp = *sp;
    }
    ol += n_o;
    if (n_o) orp = sync;
    tl += n1 + n2 + n3;
    trp = rp;
    wv_sync();
    return ok;
}
AVK_DEV bool hap_extend_seq(const Ctx &c, const HapPtr &p, HapHdr &h, bool is_truth, bool has_var, const UVar &v, u32 allele, u32 sync) {
    return is_truth ? hap_extend_seq_t<true>(c, p, h, has_var, v, allele, sync) : hap_extend_seq_t<false>(c, p, h, has_var, v, allele, sync);
}

/* DWFALite::update for a haplotype record (dynamic_wfa.rs:68-84).  While ed == 0 the wavefront is the single
 * offset h.d0: extend = slide it over the common part of the two sequences, 256 bases per step; only a real
 * mismatch (both sequences continue and differ) enters the general wavefront code. */
AVK_DEV bool hap_slide_d0(const HapPtr &p, HapHdr &h) { /* returns true when either end is touched */
    const u32 lim = h.t_len < h.q_len ? h.t_len : h.q_len;
    const u32 d = h.d0 < lim ? h.d0 + seq_match_run(p.tseq, lim, h.d0, p.qseq, lim, h.d0) : h.d0;
    h.d0 = d;
    return d >= lim;
}
/* The step from distance 0 to distance 1, in registers.  The zero-distance front stopped at offset d with both sequences
 * continuing; increase_edit_distance turns [d] into [d, d+1, d+1] (no clipping), i.e. the three diagonals start at
 *   i = 0: O offset d,   B offset d+1      i = 1: O d+1, B d+1 (the substitution)      i = 2: O d+1, B d
 * and each slides while the bases agree, 64 bases per step (the general lane-group code takes 16 per step for three
 * diagonals and keeps the front in LDS).  Leaves the front in
 * wf[0..2] and returns whether an end is touched (update's stopping rule, dynamic_wfa.rs:68-84). */
AVK_DEV bool hap_raise_to_one(const HapPtr &p, const HapHdr &h) {
    const u32 lane = (u32)wv_lane();
    const u32 d = h.d0, bl = h.t_len, ol = h.q_len;
    /* seq_match_run stops at the first difference or at either end; starts past an end give 0 */
    const u32 n0 = seq_match_run(p.tseq, bl, d + 1, p.qseq, ol, d);
    const u32 n1 = seq_match_run(p.tseq, bl, d + 1, p.qseq, ol, d + 1);
    const u32 n2 = seq_match_run(p.tseq, bl, d, p.qseq, ol, d + 1);
    const u32 o0 = d + n0, o1 = d + 1 + n1, o2 = d + 1 + n2;
    wv_sync();
    if (lane < 3) p.wf[lane] = lane == 0 ? o0 : (lane == 1 ? o1 : o2);
    wv_sync();
    /* reached_baseline_end || reached_other_end at distance 1: the offset in B is wf[i] + 1 - i */
    return o0 + 1 >= bl || o0 >= ol || o1 >= bl || o1 >= ol || o2 - 1 >= bl || o2 >= ol;
}
AVK_DEV int hap_update(const HapPtr &p, u32 wfcap, HapHdr &h) {
    if (h.ed == 0) {
        if (hap_slide_d0(p, h)) return 0;
        if (wfcap < 3) return RS_OVERFLOW; /* 2 * ed + 3 > cap at ed 0 */
        const bool touched = hap_raise_to_one(p, h);
        h.ed = 1;
        if (touched) return 0;
    }
    return dw_update(p.wf, wfcap, h.ed, p.tseq, h.t_len, p.qseq, h.q_len);
}
/* DWFALite::finalize (dynamic_wfa.rs:183-198) after an update */
AVK_DEV int hap_finalize(const HapPtr &p, u32 wfcap, HapHdr &h) {
    if (h.ed == 0) {
        if (h.d0 >= h.t_len && h.d0 >= h.q_len) return 0; /* reached_full_diagonal with ed 0 */
        wv_sync();
        st32(p.wf, h.d0);
        wv_sync();
    }
    return dw_finalize(p.wf, wfcap, h.ed, p.tseq, h.t_len, p.qseq, h.q_len);
}

/* ------------------------------------------------------------------------------------------ */
/* node pool + best-first queue                                                               */
/* ------------------------------------------------------------------------------------------ */
AVK_DEV u8 *node_at(const Ctx &c, u32 idx) { return c.pool_base + (u64)idx * c.node_bytes; }

AVK_DEV int node_alloc(Ctx &c) {
    if (c.regq) {
        if (c.freemask) {
            const int idx = avk_ctz64(c.freemask);
            c.freemask &= c.freemask - 1;
            return idx;
        }
        if (c.pool_used < c.pool_cap) return (int)(c.pool_used++);
        return RS_OVERFLOW;
    }
    if (c.nfree > 0) {
        c.nfree -= 1;
        return (int)ld32u(c.freelist + c.nfree);
    }
    if (c.pool_used < c.pool_cap) return (int)(c.pool_used++);
    return RS_OVERFLOW;
}
AVK_DEV void node_free(Ctx &c, u32 idx) {
    if (c.regq) {
        c.freemask |= 1ull << idx;
        return;
    }
    wv_sync();
    st32(c.freelist + c.nfree, idx);
    c.nfree += 1;
    wv_sync();
}
/* A haplotype record's bytes that mean something: header and allele sets, the front's 2 ed + 1 entries, the two sequences up to their lengths.  A record is laid
 * out for the window's worst case (two sequence capacities, a front of the tier's cap); in a window of kilobases that is 10-20 KB of which a node early in the
 * search uses a few hundred bytes.  Records over AVK_COPY_USED_MIN bytes are copied this way (one more round trip for the three header words, far fewer bytes). */
#ifndef AVK_COPY_USED_MIN
#define AVK_COPY_USED_MIN 2048u
#endif
AVK_DEV void hap_copy_used(const Ctx &c, u8 *dst, const u8 *src, u32 wfcap) {
    u32 w[5];
    ldvec_u<5>((const u32 *)src + H_T_LEN, w); /* t_len, q_len, t_skip, q_skip, ed: consecutive header words */
    const u32 t_len = w[0], q_len = w[1], ed = w[4];
    const u32 front = H_WORDS + 4u * c.alw; /* words: header, two allele sets */
    u32 nwf = 2u * ed + 1u;
    if (nwf > wfcap) nwf = wfcap;
    const u32 *s32 = (const u32 *)src;
    u32 *d32 = (u32 *)dst;
    copy_words(d32, s32, front + nwf);
    const u32 seq_w = front + wfcap; /* first word of tseq */
    copy_words_long(d32 + seq_w, s32 + seq_w, (t_len + 3u) >> 2);
    copy_words_long(d32 + seq_w + (c.seqcap >> 2), s32 + seq_w + (c.seqcap >> 2), (q_len + 3u) >> 2);
}
AVK_DEV void node_copy(const Ctx &c, u32 dst, u32 src, bool phase_a = true) {
    wv_sync();
    const u32 rec = phase_a ? c.hapA_bytes : c.hapB_bytes;
    if (rec < (u32)AVK_COPY_USED_MIN) copy_words((u32 *)node_at(c, dst), (const u32 *)node_at(c, src), c.node_bytes >> 2);
    else {
        u8 *d = node_at(c, dst);
        const u8 *s = node_at(c, src);
        copy_words((u32 *)d, (const u32 *)s, NODE_HDR >> 2);
        hap_copy_used(c, d + NODE_HDR, s + NODE_HDR, phase_a ? c.wfcap : 2u);
        if (phase_a) hap_copy_used(c, d + NODE_HDR + rec, s + NODE_HDR + rec, c.wfcap);
    }
    wv_sync();
}
AVK_DEV int queue_push(Ctx &c, u64 key, u32 slot) {
    if (c.qn >= c.qcap) return RS_OVERFLOW;
    if (c.regq) {
        if ((u32)wv_lane() == c.qn) {
            c.rq_key = key;
            c.rq_slot = slot;
        }
        c.qn += 1;
        return 0;
    }
    wv_sync();
    if (wv_lane() == 0) {
        c.qkeys[c.qn] = key;
        c.qslots[c.qn] = slot;
    }
    c.qn += 1;
    wv_sync();
    return 0;
}
/* pop the entry with the smallest key (keys are unique: they end in the node id) */
AVK_DEV u32 queue_pop(Ctx &c, u64 &key_out) {
    const u32 lane = (u32)wv_lane();
    if (c.regq) {
        const u64 mine = lane < c.qn ? c.rq_key : ~0ull;
        const u64 m = wv_min_u64(mine);
        const u32 pos = (u32)avk_ctz64(wv_ballot(mine == m));
        const u32 slot = wv_readlane(c.rq_slot, pos);
        const u32 last = c.qn - 1;
        const u32 lk_lo = wv_readlane((u32)c.rq_key, last), lk_hi = wv_readlane((u32)(c.rq_key >> 32), last);
        const u32 ls = wv_readlane(c.rq_slot, last);
        if (lane == pos) {
            c.rq_key = ((u64)lk_hi << 32) | lk_lo;
            c.rq_slot = ls;
        }
        c.qn = last;
        key_out = m;
        return slot;
    }
    u64 best = ~0ull;
    u32 bpos = 0xFFFFFFFFu;
    for (u32 j = lane; j < c.qn; j += 64) {
        const u64 k = c.qkeys[j];
        if (k < best) {
            best = k;
            bpos = j;
        }
    }
    const u64 m = wv_min_u64(best);
    const u32 pos = wv_min_u32(best == m ? bpos : 0xFFFFFFFFu);
    const u32 slot = ld32u(c.qslots + pos);
    const u32 last = c.qn - 1;
    wv_sync();
    if (lane == 0 && pos != last) {
        c.qkeys[pos] = c.qkeys[last];
        c.qslots[pos] = c.qslots[last];
    }
    c.qn = last;
    wv_sync();
    key_out = m;
    return slot;
}
/* register form <-> memory form (only the auto-fail filter of phase B works on the memory form) */
AVK_DEV void queue_spill(Ctx &c) {
    const u32 lane = (u32)wv_lane();
    wv_sync();
    if (lane < c.qn) {
        c.qkeys[lane] = c.rq_key;
        c.qslots[lane] = c.rq_slot;
    }
    u32 nf = 0;
    for (u64 m = c.freemask; m; m &= m - 1) {
        st32(c.freelist + nf, (u32)avk_ctz64(m));
        nf += 1;
    }
    c.nfree = nf;
    wv_sync();
}
AVK_DEV void queue_reload(Ctx &c) {
    const u32 lane = (u32)wv_lane();
    wv_sync();
    if (lane < c.qn) {
        c.rq_key = c.qkeys[lane];
        c.rq_slot = c.qslots[lane];
    }
    u64 m = 0;
    for (u32 j = 0; j < c.nfree; ++j) m |= 1ull << ld32u(c.freelist + j);
    c.freemask = m;
    c.nfree = 0;
    wv_sync();
}

/* ------------------------------------------------------------------------------------------ */
/* a region on a TEAM of wavefronts (round 6: the launch of the long windows)                  */
/* ------------------------------------------------------------------------------------------ */
/* A search in a window of kilobases is one long chain on one wavefront — 0.25 s for 92 calls on 20 kbp — but its links are not one piece of work each:
 *   - ComparisonNode::extend_variant (src/query_optimizer.rs:443-451) extends the node's two haplotypes independently (HaplotypeDWFA::extend_variant,
 *     src/dwfa/haplotype_dwfa.rs:46-67), and a heterozygous call makes two children (:269-293): up to four copy + extend + DWFA-update jobs per pop;
 *   - add_basepair_stats (src/waffle_solver.rs:335-449) is 4 alignments against the reference window and, per call type and haplotype, 2 x 2 more on
 *     filtered sequences: a dozen and more independent wfa_ed calls.
 * A workgroup of avk_region_kernel_team is ONE region's team: wave 0 (the owner) runs the search exactly as region_worker does — queue, quota, pop order, ids —
 * and POSTS those pieces as jobs in the workgroup's LDS; its three siblings (and the owner itself) claim and run them, results come back through the job records.
 * What a job reads and writes is the owner's workspace (global memory, the same CU's L1) and, for the metrics, the sibling's own scratch.  Children are made
 * OUT OF PLACE — child record = copy of the parent's record, extended — so the jobs of a pop never touch each other's bytes; the popped node is freed instead
 * of becoming the second clone.  Node contents, ids, costs and pop order are those of the one-wave search: the results are bit-identical. */
enum { TJ_HAP = 1, TJ_ED = 2, TJ_FILT = 3, TJ_GT = 4 };
enum { TJF_FINAL = 1 };
enum { TEAM_JOBS = 36 };
struct TeamJob {
    u32 kind, flags;
    u32 depth, allele; /* TJ_HAP: the call of this depth of the search order, AL_REF / AL_ALT */
    u32 x0, x1;        /* TJ_ED: lengths; TJ_FILT: side, index into the filtered types */
    u64 src, dst;      /* TJ_HAP: haplotype records (dst == src: in place); TJ_ED: the two sequences; TJ_FILT: src = the winner's haplotype record */
    int res[3];
    u32 pad_;
};
struct TeamBox {
    u32 next;   /* generation << 8 | next job to claim: the word the siblings watch */
    u32 n_jobs; /* generation << 8 | jobs of this generation */
    u32 done, quit, ver, pad_[3];
    /* what a sibling needs of the owner's Ctx (published once per region attempt, `ver` counts them) */
    u64 ws, ref, vars, alle, ovars;
    u32 L, T, Q, N, seqcap, wfcap, alw, hapA_bytes, wfs_cap, hapB_bytes, nodeB_bytes, qcap;
    u64 own_poolB; /* TJ_GT on the owner: the first node of its phase-B pool */
    u64 pad3_;
    TeamJob job[TEAM_JOBS];
};
AVK_DEV u32 gen_filtered(const Ctx &c, u32 v0, u32 cnt, const u64 *alt, u32 ftype, u8 *out, u32 &failed_ed);
AVK_DEV int phaseB(Ctx &c, const u64 *in_talt, const u64 *in_qalt, u64 *res, u32 cutoff);
AVK_DEV u32 team_sup_type(u32 s) {
    return s == 0 ? (u32)AVK_VT_SNV : (s == 1 ? (u32)AVK_VT_INSERTION : (s == 2 ? (u32)AVK_VT_DELETION : (s == 3 ? (u32)AVK_VT_INDEL : (s == 4 ? (u32)AVK_VT_TR_CONTRACTION :
           (s == 5 ? (u32)AVK_VT_TR_EXPANSION : (s == 6 ? (u32)AVK_VT_SV_DELETION : (u32)AVK_VT_SV_INSERTION))))));
}
AVK_DEV void team_publish(Ctx &c) { /* owner, after the workspace is carved */
    TeamBox *b = c.team;
    wv_sync();
    if (wv_lane() == 0) {
        b->ws = (u64)c.ws, b->ref = (u64)c.ref, b->vars = (u64)c.vars, b->alle = (u64)c.alle, b->ovars = (u64)c.ovars;
        b->L = c.L, b->T = c.T, b->Q = c.Q, b->N = c.N, b->seqcap = c.seqcap, b->wfcap = c.wfcap, b->alw = c.alw, b->hapA_bytes = c.hapA_bytes, b->wfs_cap = c.wfs_cap;
        b->hapB_bytes = c.hapB_bytes, b->nodeB_bytes = c.nodeB_bytes, b->qcap = c.qcap;
        avk_wg_store(&b->ver, avk_wg_load(&b->ver) + 1u);
    }
    wv_sync();
}
/* job j of the box, run by one wave with its own scratch (hc.wfs, hc.seq_a) on the owner's data.  A CALL, not inlined: inlined into the loops of team_run / team_helper
 * the structurizer folded the job's loops and the `lane == 0` branches behind it into one divergent loop that sent lanes 1..63 round the job again (a step that never
 * ended: tools/gpu_team_hang.py showed the wave stuck right behind its first job). */
AVK_DEV_NOINLINE void team_exec(const Ctx &hc, TeamBox *b, u32 j) {
    TeamJob *jb = b->job + j;
    const u32 kind = wv_uni(jb->kind);
    int r0 = 0, r1 = 0, r2 = 0;
    if (kind == TJ_HAP) {
        const u8 *src = (const u8 *)wv_uni64(jb->src);
        u8 *dst = (u8 *)wv_uni64(jb->dst);
        const u32 flags = wv_uni(jb->flags), depth = wv_uni(jb->depth), allele = wv_uni(jb->allele);
        if (dst != src) { /* the child's record: the parent's, copied (hap_record_copy) */
            wv_sync();
            if (hc.hapA_bytes < (u32)AVK_COPY_USED_MIN) copy_words((u32 *)dst, (const u32 *)src, hc.hapA_bytes >> 2);
            else hap_copy_used(hc, dst, src, hc.wfcap);
            wv_sync();
        }
        const HapPtr p = hap_ptr(dst, hc.alw, hc.wfcap, hc.seqcap);
        HapHdr h = hap_load(p.w);
        if (flags & TJF_FINAL) { /* nodeA_finalize */
            UVar none;
            none.rel_pos = none.a0_len = none.a1_len = none.a_off = none.raw_space = none.alt_ed = none.type = none.zyg = 0;
            hap_extend_seq(hc, p, h, true, false, none, AL_REF, hc.L);
            if (hap_update(p, hc.wfcap, h)) r0 = 1;
            else if (hap_finalize(p, hc.wfcap, h)) r0 = 1;
        } else { /* nodeA_extend, one haplotype */
            u32 sync, vi;
            const UVar v = load_ovar(hc.ovars, depth, sync, vi);
            hap_extend_seq(hc, p, h, vi < hc.T, true, v, allele, sync);
            if (hap_update(p, hc.wfcap, h)) r0 = 1;
        }
        wv_sync();
        hap_store(p.w, h);
        wv_sync();
    } else if (kind == TJ_ED) {
        r0 = wfa_ed(hc, (const u8 *)wv_uni64(jb->src), wv_uni(jb->x0), (const u8 *)wv_uni64(jb->dst), wv_uni(jb->x1));
    } else if (kind == TJ_GT) { /* optimize_gt_alleles for one haplotype of an optimum (exact_gt_optimizer.rs:108-357): a search of its own, in a pool of its own */
        Ctx lc = hc;
        u64 *me = (u64 *)wv_uni64(jb->src); /* a memo entry: [talt | qalt | result truth | result query] */
        lc.node_bytes = hc.nodeB_bytes;
        if (wv_uni(jb->x1)) { /* the owner: its own phase-B pool and queue arrays */
            lc.pool_base = (u8 *)wv_uni64(b->own_poolB);
            lc.pool_cap = wv_uni(jb->x0);
        } else { /* a sibling: queue, free list and nodes in its scratch slice */
            u8 *sc = (u8 *)hc.wfs;
            const u32 qc = hc.qcap;
            lc.qkeys = (u64 *)sc;
            lc.qslots = (u32 *)(sc + 8ull * qc);
            lc.freelist = lc.qslots + qc;
            const u64 off = AVK_ALIGN16(16ull * qc);
            lc.pool_base = sc + off;
            u64 cap = hc.team_scratch_bytes > off && hc.nodeB_bytes ? (hc.team_scratch_bytes - off) / hc.nodeB_bytes : 0;
            if (cap > qc) cap = qc;
            lc.pool_cap = (u32)cap;
        }
        r0 = phaseB(lc, me, me + hc.alw, me + 2ull * hc.alw, 0xFFFFFFFFu);
    } else { /* TJ_FILT: one side of one call type on one haplotype of the winner (waffle_solver.rs:383-445) */
        const HapPtr wp = hap_ptr((u8 *)wv_uni64(jb->src), hc.alw, hc.wfcap, hc.seqcap);
        const HapHdr hd = hap_load(wp.w);
        const u32 side = wv_uni(jb->x0), ftype = team_sup_type(wv_uni(jb->x1));
        u32 failed = 0;
        if (side) { /* query */
            const u32 fl = gen_filtered(hc, hc.T, hc.Q, wp.qalt, ftype, hc.seq_a, failed);
            r0 = wfa_ed(hc, hc.ref, hc.L, hc.seq_a, fl);
            r1 = r0 < 0 ? -1 : wfa_ed(hc, wp.tseq, hd.t_len, hc.seq_a, fl);
        } else {
            const u32 fl = gen_filtered(hc, 0, hc.T, wp.talt, ftype, hc.seq_a, failed);
            r0 = wfa_ed(hc, hc.ref, hc.L, hc.seq_a, fl);
            r1 = r0 < 0 ? -1 : wfa_ed(hc, hc.seq_a, fl, wp.qseq, hd.q_len);
        }
        r2 = (int)failed;
    }
    wv_sync();
    if (wv_lane() == 0) jb->res[0] = r0, jb->res[1] = r1, jb->res[2] = r2;
    wv_sync();
}
/* owner: the jobs 0 .. n - 1 of the box are written; every wave of the team (this one included) takes some; returns when all are done */
AVK_DEV void team_run(Ctx &c, u32 n) {
    TeamBox *b = c.team;
    const u32 lane = (u32)wv_lane();
    wv_sync();
    avk_release_wg();
    c.team_gen = (c.team_gen + 1u) & 0xFFFFFFu;
    const u32 gen = c.team_gen;
    if (lane == 0) { /* the generation opens with the LAST store: whoever sees it in `next` sees the jobs, their count and the cleared counter */
        avk_wg_store(&b->n_jobs, (gen << 8) | n);
        avk_wg_store(&b->done, 0u);
        avk_wg_store(&b->next, gen << 8);
        if (c.team_dbg) avk_st_agent_u32(c.team_dbg + 0, gen), avk_st_agent_u32(c.team_dbg + 1, 10u), avk_st_agent_u32(c.team_dbg + 2, n);
    }
    /* the jobs are dealt: job j is wave j mod 4's (every sibling is idle when a generation opens: the owner waited for the one before); with the siblings away
     * (diagnostic mode 2, the emulator) all of them are the owner's */
    const u32 stride = c.team_waves ? c.team_waves : 1u;
    u32 mine = 0;
    for (u32 j = 0; j < n; j += stride) {
        if (lane == 0 && c.team_dbg) avk_st_agent_u32(c.team_dbg + 1, 20u + j), avk_st_agent_u32(c.team_dbg + 4, b->job[j].kind);
        team_exec(c, b, j);
        mine += 1;
    }
    avk_release_wg();
    if (lane == 0) {
        (void)avk_wg_add(&b->done, mine);
        if (c.team_dbg) avk_st_agent_u32(c.team_dbg + 1, 30u);
    }
    for (;;) {
        u32 d = 0;
        if (lane == 0) d = avk_wg_load(&b->done);
        d = wv_uni(wv_shfl(d, 0));
        if (d >= n) break;
        avk_sleep_short();
    }
    avk_acquire_wg();
    wv_sync();
    if (lane == 0 && c.team_dbg) avk_st_agent_u32(c.team_dbg + 1, 40u);
}
/* a sibling wave of the team: until the owner says quit.  `scratch` = this wave's own workspace slice. */
AVK_DEV void team_helper(TeamBox *b, u8 *scratch, u64 scratch_bytes, u32 wave_in_team, u32 team_waves) {
    const u32 lane = (u32)wv_lane();
    u32 my_gen = 0, my_ver = 0;
    Ctx hc;
    hc.team = (TeamBox *)0, hc.team_gen = 0, hc.team_scratch_bytes = 0, hc.team_mode = 0, hc.team_dbg = (u32 *)0, hc.team_waves = 0;
    hc.L = hc.T = hc.Q = hc.N = hc.seqcap = hc.wfcap = hc.alw = hc.hapA_bytes = hc.wfs_cap = hc.hapB_bytes = hc.nodeB_bytes = hc.qcap = 0;
    hc.ws = (u8 *)0, hc.ref = (const u8 *)0, hc.vars = (LVar *)0, hc.alle = (u8 *)0, hc.ovars = (const AvkOrdVar *)0, hc.wfs = (u32 *)0, hc.seq_a = (u8 *)0;
    for (;;) {
        u32 v = 0, q = 0;
        for (;;) { /* a generation this wave has not seen, or the end */
            if (lane == 0) v = avk_wg_load(&b->next), q = avk_wg_load(&b->quit);
            v = wv_uni(wv_shfl(v, 0)), q = wv_uni(wv_shfl(q, 0));
            if ((v >> 8) != my_gen || q) break;
            avk_sleep_short();
        }
        if ((v >> 8) == my_gen) return; /* quit, and no generation this wave has not served */
        my_gen = v >> 8;
        u32 nj = 0, ver = 0;
        if (lane == 0) nj = avk_wg_load(&b->n_jobs), ver = avk_wg_load(&b->ver);
        nj = wv_uni(wv_shfl(nj, 0)), ver = wv_uni(wv_shfl(ver, 0));
        const u32 n = (nj >> 8) == my_gen ? (nj & 0xFFu) : 0u; /* (the count is stored before the generation opens) */
        avk_acquire_wg();
        if (ver != my_ver) { /* another region (or another attempt at it): the owner's pointers and sizes, this wave's scratch */
            my_ver = ver;
            hc.ws = (u8 *)wv_uni64(b->ws), hc.ref = (const u8 *)wv_uni64(b->ref), hc.vars = (LVar *)wv_uni64(b->vars), hc.alle = (u8 *)wv_uni64(b->alle);
            hc.ovars = (const AvkOrdVar *)wv_uni64(b->ovars);
            hc.L = wv_uni(b->L), hc.T = wv_uni(b->T), hc.Q = wv_uni(b->Q), hc.N = wv_uni(b->N), hc.seqcap = wv_uni(b->seqcap), hc.wfcap = wv_uni(b->wfcap), hc.alw = wv_uni(b->alw);
            hc.hapA_bytes = wv_uni(b->hapA_bytes), hc.wfs_cap = wv_uni(b->wfs_cap);
            hc.hapB_bytes = wv_uni(b->hapB_bytes), hc.nodeB_bytes = wv_uni(b->nodeB_bytes), hc.qcap = wv_uni(b->qcap);
            hc.team_scratch_bytes = scratch_bytes;
            hc.wfs = (u32 *)scratch;
            hc.seq_a = scratch + 4ull * hc.wfs_cap;
            /* (the owner posts metrics jobs only when 4 wfs_cap + seqcap fits: Ctx::team_scratch_bytes) */
        }
        u32 mine = 0;
        for (u32 j = wave_in_team; j < n; j += team_waves) {
            team_exec(hc, b, j);
            mine += 1;
        }
        avk_release_wg();
        if (lane == 0 && mine) (void)avk_wg_add(&b->done, mine);
    }
}
AVK_DEV void team_job_hap(TeamBox *b, u32 j, const u8 *src, u8 *dst, u32 depth, u32 allele, u32 flags) { /* lane 0 of the owner */
    TeamJob *jb = b->job + j;
    jb->kind = TJ_HAP, jb->flags = flags, jb->depth = depth, jb->allele = allele, jb->x0 = jb->x1 = 0, jb->src = (u64)src, jb->dst = (u64)dst;
}

/* ------------------------------------------------------------------------------------------ */
/* phase A — optimize_sequences (src/query_optimizer.rs:166-365)                               */
/* ------------------------------------------------------------------------------------------ */
AVK_DEV u32 nodeA_cost(const Ctx &c, u32 idx) {
    const u32 *n = (const u32 *)node_at(c, idx);
    const u32 *h0 = n + NODE_HDR / 4, *h1 = (const u32 *)((const u8 *)h0 + c.hapA_bytes);
    u32 a[3], b[3];
    ldvec_u<3>(h0 + H_T_SKIP, a); /* t_skip, q_skip, ed are consecutive header words */
    ldvec_u<3>(h1 + H_T_SKIP, b);
    return a[0] + a[1] + a[2] + b[0] + b[1] + b[2];
}

/* ComparisonNode::extend_variant (:443-451) = both haplotypes + their DWFA updates */
/* Node word 1 of a phasing-search node: 1 while its two haplotype records are identical (the root, and every node reached from
 * it by giving both haplotypes the same allele).  Extending identical records with the same allele gives identical records, so
 * the second one is copied instead of computed; and the (ALT|REF) clone of a symmetric parent is the (REF|ALT) clone with its
 * records swapped.  Pure reuse of results: ids, costs and pop order are untouched. */
AVK_DEV void hap_record_copy(const Ctx &c, u8 *dst, const u8 *src) {
    wv_sync();
    if (c.hapA_bytes < (u32)AVK_COPY_USED_MIN) copy_words((u32 *)dst, (const u32 *)src, c.hapA_bytes >> 2);
    else hap_copy_used(c, dst, src, c.wfcap);
    wv_sync();
}
AVK_DEV int nodeA_extend(const Ctx &c, u32 idx, bool is_truth, const UVar &v, u32 a1, u32 a2, u32 sync) {
    AVK_TA_DECL
    u8 *n = node_at(c, idx);
    const bool sym = ld32u((const u32 *)n + 1) != 0;
    const bool twin = sym && a1 == a2; /* the second record will equal the first */
    for (int hh = 0; hh < (twin ? 1 : 2); ++hh) {
        const HapPtr p = hap_ptr(n + NODE_HDR + (u64)hh * c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
        HapHdr h = hap_load(p.w);
        AVK_TA_MARK(const_cast<Ctx &>(c), 12)
        hap_extend_seq(c, p, h, is_truth, true, v, hh == 0 ? a1 : a2, sync);
        AVK_TA_MARK(const_cast<Ctx &>(c), 14)
        if (hap_update(p, c.wfcap, h)) return RS_OVERFLOW;
        AVK_TA_MARK(const_cast<Ctx &>(c), 15)
        wv_sync();
        hap_store(p.w, h);
        wv_sync();
    }
    if (twin) hap_record_copy(c, n + NODE_HDR + c.hapA_bytes, n + NODE_HDR);
    else if (sym) {
        st32((u32 *)n + 1, 0);
        wv_sync();
    }
    return 0;
}
/* node `idx` := node `from` with its two haplotype records swapped (and no longer symmetric) */
AVK_DEV void nodeA_mirror(const Ctx &c, u32 idx, u32 from) {
    u8 *n = node_at(c, idx);
    const u8 *f = node_at(c, from);
    hap_record_copy(c, n + NODE_HDR, f + NODE_HDR + c.hapA_bytes);
    hap_record_copy(c, n + NODE_HDR + c.hapA_bytes, f + NODE_HDR);
    st32((u32 *)n + 1, 0);
    wv_sync();
}

/* ComparisonNode::finalize_dwfas (:457-462, haplotype_dwfa.rs:84-95) */
AVK_DEV int nodeA_finalize(const Ctx &c, u32 idx) {
    u8 *n = node_at(c, idx);
    UVar none;
    none.rel_pos = none.a0_len = none.a1_len = none.a_off = none.raw_space = none.alt_ed = none.type = none.zyg = 0;
    const bool twin = ld32u((const u32 *)n + 1) != 0; /* identical haplotype records: finalise one, copy it (nodeA_extend) */
    for (int hh = 0; hh < (twin ? 1 : 2); ++hh) {
        const HapPtr p = hap_ptr(n + NODE_HDR + (u64)hh * c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
        HapHdr h = hap_load(p.w);
        hap_extend_seq(c, p, h, true, false, none, AL_REF, c.L); /* both sides to the region end */
        if (hap_update(p, c.wfcap, h)) return RS_OVERFLOW;
        if (hap_finalize(p, c.wfcap, h)) return RS_OVERFLOW;
        wv_sync();
        hap_store(p.w, h);
        wv_sync();
    }
    if (twin) hap_record_copy(c, n + NODE_HDR + c.hapA_bytes, n + NODE_HDR);
    return 0;
}

/* returns the number of tied optima (their node indices are in c.optlist), RS_OVERFLOW, or -status-100 */
template <bool TEAM = false> AVK_DEV int phaseA(Ctx &c) {
    AVK_TA_DECL
    c.node_bytes = c.nodeA_bytes;
    c.pool_base = c.pool;
    c.pool_cap = (u32)(c.pool_bytes / c.nodeA_bytes);
    if (c.pool_cap > c.qcap) c.pool_cap = c.qcap;
    c.pool_used = 0;
    c.nfree = 0;
    c.qn = 0;
    if (c.pool_cap < 3) return RS_OVERFLOW;
    c.regq = c.pool_cap <= 64;
    c.regb = c.N < 64;
    c.freemask = 0;
    c.rq_key = 0;
    c.rq_slot = 0;
    c.rbucket = 0;
    if (!c.regb) zero_words(c.bucket, c.N + 1);

    int root = node_alloc(c);
    {
        u8 *n = node_at(c, (u32)root);
        st32((u32 *)n, 0);
        st32((u32 *)n + 1, 1); /* the two haplotypes of the root are identical (nodeA_extend) */
        hap_init(hap_ptr(n + NODE_HDR, c.alw, c.wfcap, c.seqcap), c.alw);
        hap_init(hap_ptr(n + NODE_HDR + c.hapA_bytes, c.alw, c.wfcap, c.seqcap), c.alw);
    }
    wv_sync();
    queue_push(c, 0, (u32)root);
    u32 next_id = 1;
    u32 best_ed = 0xFFFFFFFFu;
    u32 nbest = 0;
    AVK_TA_MARK(c, 8)

    while (c.qn > 0) {
        u64 key;
        const u32 ni = queue_pop(c, key);
        const u32 cost = (u32)(key >> 32);
        if (cost > best_ed) break; /* :204 skips it — and, pops being in non-decreasing cost order (a child never
                                      costs less than its parent), every node still queued would be skipped too */
        const u32 *nw = (const u32 *)node_at(c, ni);
        const u32 depth = ld32u(nw + NODE_HDR / 4 + H_T_NAL) + ld32u(nw + NODE_HDR / 4 + H_Q_NAL); /* set_alleles of hap 1, :478-481 */
        const u32 cnt = c.regb ? wv_readlane(c.rbucket, depth) : ld32u(c.bucket + depth);
        if (cnt >= c.max_branch) { /* :222 */
            node_free(c, ni);
            continue;
        }
        if (c.regb) {
            if ((u32)wv_lane() == depth) c.rbucket += 1;
        } else {
            wv_sync();
            st32(c.bucket + depth, cnt + 1);
            wv_sync();
        }

        AVK_TA_MARK(c, 9)
        if (depth == c.N) { /* :227-247 */
            if (TEAM && c.team) { /* the node's two haplotypes side by side (a twin's second record is computed like the first) */
                u8 *n = node_at(c, ni);
                wv_sync();
                if (wv_lane() == 0) {
                    team_job_hap(c.team, 0, n + NODE_HDR, n + NODE_HDR, 0, AL_REF, TJF_FINAL);
                    team_job_hap(c.team, 1, n + NODE_HDR + c.hapA_bytes, n + NODE_HDR + c.hapA_bytes, 0, AL_REF, TJF_FINAL);
                }
                team_run(c, 2);
                if (wv_uni((u32)c.team->job[0].res[0]) | wv_uni((u32)c.team->job[1].res[0])) return RS_OVERFLOW;
            } else if (nodeA_finalize(c, ni)) return RS_OVERFLOW;
            const u32 fc = nodeA_cost(c, ni);
            if (fc < best_ed) {
                for (u32 k = 0; k < nbest; ++k) node_free(c, ld32u(c.optlist + k));
                best_ed = fc;
                nbest = 0;
            }
            if (fc == best_ed) {
                if (nbest >= c.optcap) return RS_OVERFLOW;
                wv_sync();
                st32(c.optlist + nbest, ni);
                wv_sync();
                nbest += 1;
            } else {
                node_free(c, ni);
            }
            AVK_TA_MARK(c, 10)
            continue;
        }

        u32 sync, vi; /* sync: position of the next variant, or the window end (:258-265) */
        const UVar v = load_ovar(c.ovars, depth, sync, vi);
        const bool is_truth = vi < c.T;
        const u32 zyg = v.zyg;
        const bool het = zyg == AVK_ZYG_UNPHASED_HET || zyg == AVK_ZYG_PHASED_HET01 || zyg == AVK_ZYG_PHASED_HET10;

        if (TEAM && c.team) {
            const bool two = het && (!is_truth || zyg == AVK_ZYG_UNPHASED_HET);
            u8 *pn = node_at(c, ni);
            const bool sym = ld32u((const u32 *)pn + 1) != 0;
            if (two) { /* :269-293: two clones, (REF|ALT) then (ALT|REF) — both made out of place from the popped node, four records on four waves */
                const int c1 = node_alloc(c);
                if (c1 < 0) return RS_OVERFLOW;
                const int c2 = node_alloc(c); /* (a full pool: the popped node becomes the second clone in place, behind the first one's jobs — what the one-wave search does) */
                u8 *n1 = node_at(c, (u32)c1), *n2 = c2 >= 0 ? node_at(c, (u32)c2) : pn;
                wv_sync();
                if (wv_lane() == 0) {
                    ((u32 *)n1)[0] = next_id, ((u32 *)n1)[1] = 0, ((u32 *)n1)[2] = ((const u32 *)pn)[2], ((u32 *)n1)[3] = ((const u32 *)pn)[3];
                    if (c2 >= 0) ((u32 *)n2)[0] = next_id + 1, ((u32 *)n2)[1] = 0, ((u32 *)n2)[2] = ((const u32 *)pn)[2], ((u32 *)n2)[3] = ((const u32 *)pn)[3];
                    team_job_hap(c.team, 0, pn + NODE_HDR, n1 + NODE_HDR, depth, AL_REF, 0);
                    team_job_hap(c.team, 1, pn + NODE_HDR + c.hapA_bytes, n1 + NODE_HDR + c.hapA_bytes, depth, AL_ALT, 0);
                    if (c2 >= 0) {
                        team_job_hap(c.team, 2, pn + NODE_HDR, n2 + NODE_HDR, depth, AL_ALT, 0);
                        team_job_hap(c.team, 3, pn + NODE_HDR + c.hapA_bytes, n2 + NODE_HDR + c.hapA_bytes, depth, AL_REF, 0);
                    }
                }
                team_run(c, c2 >= 0 ? 4u : 2u);
                u32 bad = 0;
                for (int j = 0; j < (c2 >= 0 ? 4 : 2); ++j) bad |= wv_uni((u32)c.team->job[j].res[0]);
                if (c2 < 0) {
                    wv_sync();
                    if (wv_lane() == 0) {
                        ((u32 *)pn)[0] = next_id + 1, ((u32 *)pn)[1] = 0;
                        team_job_hap(c.team, 0, pn + NODE_HDR, pn + NODE_HDR, depth, AL_ALT, 0);
                        team_job_hap(c.team, 1, pn + NODE_HDR + c.hapA_bytes, pn + NODE_HDR + c.hapA_bytes, depth, AL_REF, 0);
                    }
                    team_run(c, 2);
                    bad |= wv_uni((u32)c.team->job[0].res[0]) | wv_uni((u32)c.team->job[1].res[0]);
                }
                if (bad) return RS_OVERFLOW;
                const u32 second = c2 >= 0 ? (u32)c2 : ni;
                if (queue_push(c, ((u64)nodeA_cost(c, (u32)c1) << 32) | next_id, (u32)c1)) return RS_OVERFLOW;
                if (queue_push(c, ((u64)nodeA_cost(c, second) << 32) | (next_id + 1), second)) return RS_OVERFLOW;
                next_id += 2;
                if (c2 >= 0) node_free(c, ni);
            } else { /* :294-327: the node is moved, its id kept: its two haplotypes in place on two waves */
                u32 a1 = AL_ALT, a2 = AL_ALT;
                if (het) {
                    a1 = zyg == AVK_ZYG_PHASED_HET01 ? AL_REF : AL_ALT;
                    a2 = zyg == AVK_ZYG_PHASED_HET01 ? AL_ALT : AL_REF;
                }
                const u32 id = ld32u((const u32 *)pn);
                wv_sync();
                if (wv_lane() == 0) {
                    team_job_hap(c.team, 0, pn + NODE_HDR, pn + NODE_HDR, depth, a1, 0);
                    team_job_hap(c.team, 1, pn + NODE_HDR + c.hapA_bytes, pn + NODE_HDR + c.hapA_bytes, depth, a2, 0);
                    if (sym && a1 != a2) ((u32 *)pn)[1] = 0; /* no longer two identical records (nodeA_extend) */
                }
                team_run(c, 2);
                if (wv_uni((u32)c.team->job[0].res[0]) | wv_uni((u32)c.team->job[1].res[0])) return RS_OVERFLOW;
                if (queue_push(c, ((u64)nodeA_cost(c, ni) << 32) | id, ni)) return RS_OVERFLOW;
            }
        } else if (het && (!is_truth || zyg == AVK_ZYG_UNPHASED_HET)) { /* :269-293: two clones, (REF|ALT) then (ALT|REF) */
            const int c1 = node_alloc(c);
            if (c1 < 0) return RS_OVERFLOW;
            node_copy(c, (u32)c1, ni);
            st32((u32 *)node_at(c, (u32)c1), next_id);
            wv_sync();
            AVK_TA_MARK(c, 11)
            if (nodeA_extend(c, (u32)c1, is_truth, v, AL_REF, AL_ALT, sync)) return RS_OVERFLOW;
            AVK_TA_MARK(c, 12)
            if (queue_push(c, ((u64)nodeA_cost(c, (u32)c1) << 32) | next_id, (u32)c1)) return RS_OVERFLOW;
            next_id += 1;
            /* the popped node itself becomes the second clone */
            const bool parent_sym = ld32u((const u32 *)node_at(c, ni) + 1) != 0;
            st32((u32 *)node_at(c, ni), next_id);
            wv_sync();
            AVK_TA_MARK(c, 13)
            if (parent_sym) nodeA_mirror(c, ni, (u32)c1); /* (ALT|REF) of identical haplotypes = (REF|ALT) swapped */
            else if (nodeA_extend(c, ni, is_truth, v, AL_ALT, AL_REF, sync)) return RS_OVERFLOW;
            AVK_TA_MARK(c, 12)
            if (queue_push(c, ((u64)nodeA_cost(c, ni) << 32) | next_id, ni)) return RS_OVERFLOW;
            next_id += 1;
            AVK_TA_MARK(c, 13)
        } else { /* :294-327: the node is moved, its id kept */
            u32 a1 = AL_ALT, a2 = AL_ALT;
            if (het) {
                a1 = zyg == AVK_ZYG_PHASED_HET01 ? AL_REF : AL_ALT;
                a2 = zyg == AVK_ZYG_PHASED_HET01 ? AL_ALT : AL_REF;
            }
            const u32 id = ld32u((const u32 *)node_at(c, ni));
            AVK_TA_MARK(c, 11)
            if (nodeA_extend(c, ni, is_truth, v, a1, a2, sync)) return RS_OVERFLOW;
            AVK_TA_MARK(c, 12)
            if (queue_push(c, ((u64)nodeA_cost(c, ni) << 32) | id, ni)) return RS_OVERFLOW;
            AVK_TA_MARK(c, 13)
        }
    }
    if (nbest == 0) return -100 - AVK_ST_NO_RESULTS; /* :331 */
    c.best_cost = best_ed;
    return (int)nbest;
}

/* ------------------------------------------------------------------------------------------ */
/* phase B — optimize_gt_alleles (src/exact_gt_optimizer.rs:108-357)                           */
/* The DWFA of an ExactMatchNode has max_edit_distance 0 (:380): a live node's two sequences    */
/* agree on their common prefix, so the whole wavefront is the single value d = matched length. */
/* ------------------------------------------------------------------------------------------ */
/* ExactMatchNode::extend_variant (:395-414): returns success && is_exact_match.  With max_edit_distance 0 the
 * update either slides d0 to an end (still exact) or would have to raise the distance (not exact). */
AVK_DEV bool nodeB_extend(const Ctx &c, u32 idx, bool is_truth, const UVar &v, u32 allele, u32 sync, bool is_error) {
    u8 *n = node_at(c, idx);
    const HapPtr p = hap_ptr(n + NODE_HDR, c.alw, 2, c.seqcap);
    HapHdr h = hap_load(p.w);
    const bool ok = hap_extend_seq(c, p, h, is_truth, true, v, allele, sync);
    const bool exact = hap_slide_d0(p, h);
    if (is_error) {
        const u32 e = ld32u((u32 *)n + 1);
        wv_sync();
        st32((u32 *)n + 1, e + 1);
    }
    wv_sync();
    hap_store(p.w, h);
    wv_sync();
    return ok && exact;
}

AVK_DEV u64 keyB(u32 errors, u32 depth, u32 id) { /* (Reverse(errors), set - errors, Reverse(id)), :452-458 */
    return ((u64)errors << 48) | ((u64)(0xFFFFu - (depth - errors)) << 32) | id;
}

/* runs one haplotype with the input alleles in_talt / in_qalt.
 * result: errors (>= 0) and the final allele bit-sets in res[0..alw) truth, res[alw..2alw) query.
 * Pops come in non-decreasing error order (a child never has fewer errors than its parent), so
 *   - the first finalised exact node is the answer: everything popped after it is skipped by :169;
 *   - once a popped node has `cutoff` errors the answer is >= cutoff and the caller no longer needs it
 *     (only meaningful for N < 500, where the search cannot fail later on: the all-REF chain finishes
 *     within N expansions of any auto-fail, :309-339).
 * returns errors, RS_CUT, RS_OVERFLOW, or -status-100 */
AVK_DEV int phaseB(Ctx &c, const u64 *in_talt, const u64 *in_qalt, u64 *res, u32 cutoff) {
    c.qn = 0;
    c.pool_used = 0;
    c.nfree = 0;
    if (c.pool_cap < 3) return RS_OVERFLOW;
    c.regq = c.pool_cap <= 64;
    c.freemask = 0;
    c.rq_key = 0;
    c.rq_slot = 0;
    int root = node_alloc(c);
    {
        u8 *n = node_at(c, (u32)root);
        st32((u32 *)n, 0);
        st32((u32 *)n + 1, 0);
        hap_init(hap_ptr(n + NODE_HDR, c.alw, 2, c.seqcap), c.alw);
    }
    wv_sync();
    queue_push(c, keyB(0, 0, 0), (u32)root);
    u32 next_id = 1;
    u32 min_sync = 0, af_index = 0, af_counts = 0;

    while (c.qn > 0) {
        u64 key;
        const u32 ni = queue_pop(c, key);
        u8 *n = node_at(c, ni);
        const u32 errors = (u32)(key >> 48);
        if (errors >= cutoff) return RS_CUT;
        const HapPtr p = hap_ptr(n + NODE_HDR, c.alw, 2, c.seqcap);
        HapHdr h = hap_load(p.w);
        const u32 depth = h.t_nal + h.q_nal;
        if (depth == c.N) { /* :180-192 finalize: both to the region end, exact iff identical */
            UVar none;
            none.rel_pos = none.a0_len = none.a1_len = none.a_off = none.raw_space = none.alt_ed = none.type = none.zyg = 0;
            hap_extend_seq(c, p, h, true, false, none, AL_REF, c.L);
            const bool touched = hap_slide_d0(p, h);
            const bool exact = touched && h.d0 >= h.t_len && h.d0 >= h.q_len;
            if (exact) { /* :187-190; later pops cannot improve on it */
                wv_sync();
                for (u32 k = (u32)wv_lane(); k < c.alw; k += 64) {
                    res[k] = p.talt[k];
                    res[c.alw + k] = p.qalt[k];
                }
                wv_sync();
                return (int)errors;
            }
            node_free(c, ni);
            continue;
        }
        if (depth < min_sync) { /* :194-197 */
            node_free(c, ni);
            continue;
        }
        /* is_synchronized (haplotype_dwfa.rs:99-112); ed == 0 for every queued node */
        if (h.t_len == h.q_len && h.t_refpos == h.q_refpos) { /* :206-217 */
            min_sync = depth;
            af_counts = 0;
            af_index = min_sync;
        }
        u32 sync, vi;
        const UVar v = load_ovar(c.ovars, depth, sync, vi);
        const bool is_truth = vi < c.T;
        const u32 sub = is_truth ? vi : vi - c.T;
        const u64 *in = is_truth ? in_talt : in_qalt;
        const bool cur_alt = (wv_uni((u32)((in[sub >> 6] >> (sub & 63)) & 1))) != 0;
        const u32 id = ld32u((const u32 *)n);

        if (!cur_alt) { /* :257-273 */
            if (nodeB_extend(c, ni, is_truth, v, AL_REF, sync, false)) {
                if (queue_push(c, keyB(errors, depth + 1, id), ni)) return RS_OVERFLOW;
            } else node_free(c, ni);
        } else { /* :274-306: (REF, error) first, then ALT unless auto-failed */
            const bool do_alt = !(depth < af_index);
            u32 ref_node = ni;
            if (do_alt) { /* clone for the REF child, keep the popped node for the ALT child */
                const int c1 = node_alloc(c);
                if (c1 < 0) return RS_OVERFLOW;
                node_copy(c, (u32)c1, ni, false);
                ref_node = (u32)c1;
            }
            st32((u32 *)node_at(c, ref_node), next_id);
            wv_sync();
            if (nodeB_extend(c, ref_node, is_truth, v, AL_REF, sync, true)) {
                if (queue_push(c, keyB(errors + 1, depth + 1, next_id), ref_node)) return RS_OVERFLOW;
            } else node_free(c, ref_node);
            next_id += 1;
            if (do_alt) {
                st32((u32 *)node_at(c, ni), next_id);
                wv_sync();
                if (nodeB_extend(c, ni, is_truth, v, AL_ALT, sync, false)) {
                    if (queue_push(c, keyB(errors, depth + 1, next_id), ni)) return RS_OVERFLOW;
                } else node_free(c, ni);
                next_id += 1;
            }
        }

        af_counts += 1; /* :309-339 */
        if (af_counts >= 500) {
            if (af_index >= c.N) return -100 - AVK_ST_AUTOFAIL_OOB;
            const bool was_reg = c.regq;
            if (was_reg) {
                queue_spill(c);
                c.regq = false;
            }
            const u32 fi = ld32u(&c.ovars[af_index].vi);
            const bool f_truth = fi < c.T;
            const u32 fsub = f_truth ? fi : fi - c.T;
            const u32 lane = (u32)wv_lane();
            u32 kept = 0;
            const u32 qn0 = c.qn;
            for (u32 base = 0; base < qn0; base += 64) {
                const u32 j = base + lane;
                bool keep = false, drop = false;
                u64 k = 0;
                u32 s = 0;
                if (j < qn0) {
                    k = c.qkeys[j];
                    s = c.qslots[j];
                    const u8 *nn = node_at(c, s);
                    const HapPtr q = hap_ptr((u8 *)nn + NODE_HDR, c.alw, 2, c.seqcap);
                    const u32 nal = f_truth ? q.w[H_T_NAL] : q.w[H_Q_NAL];
                    const u64 *bits = f_truth ? q.talt : q.qalt;
                    const bool is_alt = fsub < nal && ((bits[fsub >> 6] >> (fsub & 63)) & 1);
                    keep = !is_alt;
                    drop = is_alt;
                }
                const u64 km = wv_ballot(keep), dm = wv_ballot(drop);
                const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
                wv_sync();
                if (keep) {
                    const u32 dst = kept + (u32)avk_popc64(km & below);
                    c.qkeys[dst] = k;
                    c.qslots[dst] = s;
                }
                if (drop) c.freelist[c.nfree + (u32)avk_popc64(dm & below)] = s;
                kept += (u32)avk_popc64(km);
                c.nfree += (u32)avk_popc64(dm);
                wv_sync();
            }
            c.qn = kept;
            if (was_reg) {
                c.regq = true;
                queue_reload(c);
            }
            af_index += 1;
            af_counts = 0;
        }
    }
    return -100 - AVK_ST_NO_GT_RESULT; /* :345-348 */
}

/* ------------------------------------------------------------------------------------------ */
/* phase C — metrics (src/waffle_solver.rs:296-522, src/data_types/grouped_metrics.rs:183-277)  */
/* ------------------------------------------------------------------------------------------ */
AVK_DEV u32 *gfield(u32 *g, int group, int field) { return g + group * AVK_N_FIELDS + field; }

/* GroupMetrics::add_truth_zygosity (grouped_metrics.rs:183-227) for one variant into joint + type
 * group.  Query variants are scored "as truth" and then moved to the query columns by
 * add_swap_benchmark (grouped_metrics.rs:268-277: query_tp <- truth_tp, query_fp <- truth_fn,
 * query_fp_gt <- truth_fn_gt); `q` selects those columns directly. */
/* Only the joint group and the groups of the variant types that occur in the region can hold anything: `gmask` has bit 0
 * and bit 1 + t for every such type.  Two groups per step: lanes 0..21 take one, lanes 32..53 the next.  Returns through
 * `idx` the metric-block index this lane handles in the current step, or AVK_N_GROUPS * AVK_N_FIELDS when it has none. */
AVK_DEV u32 gm_step(u32 &left, u32 lane) {
    const u32 ga = (u32)avk_ctz64(left);
    left &= left - 1;
    u32 gb = 0xFFu;
    if (left) {
        gb = (u32)avk_ctz64(left);
        left &= left - 1;
    }
    const u32 g = (lane >> 5) ? gb : ga, f = lane & 31u;
    return (g != 0xFFu && f < AVK_N_FIELDS) ? g * AVK_N_FIELDS + f : (u32)(AVK_N_GROUPS * AVK_N_FIELDS);
}
AVK_DEV void gm_zero(u32 *gm, u32 gmask) {
    const u32 lane = (u32)wv_lane();
    for (u32 left = gmask; left;) {
        const u32 i = gm_step(left, lane);
        if (i < AVK_N_GROUPS * AVK_N_FIELDS) gm[i] = 0;
    }
}
AVK_DEV void gm_add(u32 *g, bool q, u32 type, u32 w, u32 exp, u32 obs) {
    const int f_gt_tp = q ? AVK_F_GT_QUERY_TP : AVK_F_GT_TRUTH_TP, f_gt_fn = q ? AVK_F_GT_QUERY_FP : AVK_F_GT_TRUTH_FN;
    const int f_gt_fn_gt = q ? AVK_F_GT_QUERY_FP_GT : AVK_F_GT_TRUTH_FN_GT;
    const int f_hap_tp = q ? AVK_F_HAP_QUERY_TP : AVK_F_HAP_TRUTH_TP, f_hap_fn = q ? AVK_F_HAP_QUERY_FP : AVK_F_HAP_TRUTH_FN;
    const int f_w_tp = q ? AVK_F_WHAP_QUERY_TP : AVK_F_WHAP_TRUTH_TP, f_w_fn = q ? AVK_F_WHAP_QUERY_FP : AVK_F_WHAP_TRUTH_FN;
    for (int pass = 0; pass < 2; ++pass) {
        const int grp = pass == 0 ? 0 : 1 + (int)type;
        if (exp == obs) {
            avk_atomic_add_u32(gfield(g, grp, f_hap_tp), exp);
            avk_atomic_add_u32(gfield(g, grp, f_w_tp), exp * w);
            avk_atomic_add_u32(gfield(g, grp, f_gt_tp), 1);
        } else {
            avk_atomic_add_u32(gfield(g, grp, f_hap_tp), obs);
            avk_atomic_add_u32(gfield(g, grp, f_hap_fn), exp - obs);
            avk_atomic_add_u32(gfield(g, grp, f_w_tp), obs * w);
            avk_atomic_add_u32(gfield(g, grp, f_w_fn), (exp - obs) * w);
            avk_atomic_add_u32(gfield(g, grp, f_gt_fn), 1);
            if (obs > 0) avk_atomic_add_u32(gfield(g, grp, f_gt_fn_gt), 1);
        }
    }
}

/* generate_allele_sequence (waffle_solver.rs:726-778) for the variants of one side that have
 * type `ftype`, alleles taken from the optimum's bit-set.  returns the length; failed_ed out */
AVK_DEV u32 gen_filtered(const Ctx &c, u32 v0, u32 cnt, const u64 *alt, u32 ftype, u8 *out, u32 &failed_ed) {
    u32 cur = 0, len = 0, failed = 0;
    for (u32 k = 0; k < cnt; ++k) {
        const UVar v = load_var(c.vars, v0 + k);
        if (v.type != ftype) continue;
        const bool is_alt = wv_uni((u32)((alt[k >> 6] >> (k & 63)) & 1)) != 0;
        if (!is_alt) continue; /* :738-741 */
        const u32 vpos = v.rel_pos;
        if (vpos < cur) { /* :745-753 */
            failed += v.alt_ed;
            continue;
        }
        copy_bytes(out + len, c.ref + cur, vpos - cur);
        len += vpos - cur;
        copy_bytes(out + len, c.alle + v.a_off + v.a0_len, v.a1_len);
        len += v.a1_len;
        cur = vpos + v.a0_len;
    }
    if (cur < c.L) {
        copy_bytes(out + len, c.ref + cur, c.L - cur);
        len += c.L - cur;
    }
    failed_ed = failed;
    wv_sync();
    return len;
}

/* generate_exact_match (waffle_solver.rs:534-601), the hidden --enable-exact-shortcut: taken when the
 * first optimum is an exact match (all tied optima share the total cost, so cost 0 decides).  Counts come
 * from the INPUT zygosities, every variant is a TP, BASEPAIR is the distance of the (identical) haplotypes
 * to the reference plus per-variant allele distances; no RECORD_BP, no entries for absent variant types. */
AVK_DEV int exact_shortcut_metrics(const AvkKernelArgs &a, Ctx &c, u32 v_off, const HapPtr &w0, const HapHdr &h0, const HapPtr &w1,
                                   const HapHdr &h1, u32 &present_out) {
    const u32 lane = (u32)wv_lane();
    zero_words(c.gm, AVK_N_GROUPS * AVK_N_FIELDS);
    wv_sync();
    u32 l_present = 0, l_bad = 0;
    for (u32 k = lane; k < c.N; k += 64) {
        const bool is_truth = k < c.T;
        const LVar v = c.vars[k];
        const u32 vtype = v.type_zyg & 0xFF, z = (v.type_zyg >> 8) & 0xFF;
        const u32 ev = z == AVK_ZYG_HOM_ALT ? 2u : 1u; /* Unknown / HomRef never get here (pre_status) */
        for (int pass = 0; pass < 2; ++pass) {
            u32 *g = c.gm + (pass == 0 ? 0 : 1 + vtype) * AVK_N_FIELDS;
            avk_atomic_add_u32(g + (is_truth ? AVK_F_GT_TRUTH_TP : AVK_F_GT_QUERY_TP), 1);
            avk_atomic_add_u32(g + (is_truth ? AVK_F_HAP_TRUTH_TP : AVK_F_HAP_QUERY_TP), ev);
            avk_atomic_add_u32(g + (is_truth ? AVK_F_WHAP_TRUTH_TP : AVK_F_WHAP_QUERY_TP), ev * v.alt_ed);
        }
        avk_atomic_add_u32(c.gm + (1 + vtype) * AVK_N_FIELDS + (is_truth ? AVK_F_BP_TRUTH_TP : AVK_F_BP_QUERY_TP), ev * 2 * v.alt_ed);
        l_present |= 1u << vtype;
        const u32 b0 = (u32)(((is_truth ? w0.talt : w0.qalt)[(is_truth ? k : k - c.T) >> 6] >> ((is_truth ? k : k - c.T) & 63)) & 1);
        const u32 b1 = (u32)(((is_truth ? w1.talt : w1.qalt)[(is_truth ? k : k - c.T) >> 6] >> ((is_truth ? k : k - c.T) & 63)) & 1);
        if (b0 + b1 == 0) l_bad = AVK_ST_BAD_ZYGOSITY;
        const u32 rz = b0 && b1 ? AVK_ZYG_HOM_ALT : (b0 ? AVK_ZYG_PHASED_HET10 : AVK_ZYG_PHASED_HET01);
        a.var_out[v_off + k] = ev | (ev << 8) | ((u32)AVK_CLASS_TP << 16) | (rz << 24);
    }
    if (wv_max_u32(l_bad)) return AVK_ST_BAD_ZYGOSITY;
    u32 present = 0;
    for (int t = 0; t < AVK_N_VARIANT_TYPES; ++t)
        if (wv_ballot((l_present >> t) & 1)) present |= 1u << t;
    /* truth == query on both haplotypes (asserted at :561-562; cost 0 guarantees it) */
    const int e1 = wfa_ed(c, c.ref, c.L, w0.tseq, h0.t_len);
    if (e1 < 0) return RS_OVERFLOW;
    const int e2 = wfa_ed(c, c.ref, c.L, w1.tseq, h1.t_len);
    if (e2 < 0) return RS_OVERFLOW;
    wv_sync();
    if (lane == 0) {
        c.gm[AVK_F_BP_TRUTH_TP] += 2u * (u32)(e1 + e2);
        c.gm[AVK_F_BP_QUERY_TP] += 2u * (u32)(e1 + e2);
    }
    wv_sync();
    present_out = present;
    return AVK_ST_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* one region on one workspace tier                                                           */
/* ------------------------------------------------------------------------------------------ */
struct RegionOut {
    u32 ed1, ed2, n_opt, present;
};

/* returns AVK_ST_* (>= 0) or RS_OVERFLOW */
template <bool TEAM = false> AVK_DEV int solve_region_tier(const AvkKernelArgs &a, u32 r, u8 *ws, u64 ws_bytes, u32 ed_cap, Ctx &c, RegionOut &out, u32 &winner_node) {
    AVK_T_DECL
    const AvkDevRegion reg = a.regions[r];
    const u32 lane = (u32)wv_lane();
    c.L = wv_uni(reg.len);
    c.T = wv_uni(reg.t_cnt);
    c.Q = wv_uni(reg.q_cnt);
    c.N = c.T + c.Q;
    c.max_branch = a.max_branch_factor;
    const u32 v_off = wv_uni(reg.v_off);

    /* sizes come with the region record (the host packer computed them) */
    const u32 alle_bytes = wv_uni(reg.alle_bytes);
    const u32 blob_bytes = wv_uni(reg.blob_bytes);
    c.seqcap = (c.L + wv_uni(reg.grow) + 7u) & ~7u;
    if (c.seqcap == 0) c.seqcap = 8;
    const u32 maxT = c.T > c.Q ? c.T : c.Q;
    c.alw = maxT ? (maxT + 63) >> 6 : 1;
    const u32 wf_full = 2 * c.seqcap + 4;
    /* LDS tiers: the region's own bound when it is lower than the tier's cap (AvkDevRegion::ed_bound: the sum of its calls' edit distances — no haplotype
     * pair of the region can be further apart), else the tier's cap, beyond which the region goes on to the next tier.  HBM per-wave slices
     * (AVK_CAP_BOUND_ONLY): the region's own bound when the tier's cap covers it, otherwise NO cap — a region is never sent on to the few shared big slices
     * for its wavefronts.  Either way the front gets 2 cap + 2 entries instead of two per base of the window: a node of a 3 kbp window is 20 KB instead of 100. */
    u32 cap = ed_cap & ~(u32)AVK_CAP_BOUND_ONLY;
    if (cap) {
        const u32 b = wv_uni(reg.ed_bound);
        if (b < cap) cap = b ? b : 1u;
        else if (ed_cap & (u32)AVK_CAP_BOUND_ONLY) cap = 0;
    }
    c.wfcap = cap ? (2 * cap + 2) : wf_full; /* even */
    if (c.wfcap > wf_full) c.wfcap = wf_full;
    c.wfs_cap = c.wfcap;
    c.hapA_bytes = (u32)AVK_ALIGN16(H_WORDS * 4 + 16 * c.alw + 4 * c.wfcap + 2 * c.seqcap);
    c.nodeA_bytes = NODE_HDR + 2 * c.hapA_bytes;
    c.hapB_bytes = (u32)AVK_ALIGN16(H_WORDS * 4 + 16 * c.alw + 4 * 2 + 2 * c.seqcap);
    c.nodeB_bytes = NODE_HDR + c.hapB_bytes;
    c.optcap = c.max_branch < 4096 ? c.max_branch : 4096;

    /* carve the workspace */
    c.ws = ws;
    u64 off = 0;
    u8 *refbuf = ws + off;
    off = AVK_ALIGN16(off + c.L);
    /* the region blob lands here as it is: variant records | allele bytes | variants in search order | per-type counts
     * (AvkBlobVar in avk_dev_types.h; every section padded to 16 bytes) */
    u32 *const blob_dst = (u32 *)(ws + off);
    c.vars = (LVar *)(ws + off);
    off = AVK_ALIGN16(off + (u64)c.N * sizeof(LVar));
    c.alle = ws + off;
    off = AVK_ALIGN16(off + alle_bytes);
    c.ovars = (const AvkOrdVar *)(ws + off);
    off += (u64)c.N * sizeof(AvkOrdVar);
    c.counts = (u32 *)(ws + off);
    off += 32;
    c.bucket = (u32 *)(ws + off);
    off = AVK_ALIGN8(off + 4ull * (c.N + 1));
    c.optlist = (u32 *)(ws + off);
    off = AVK_ALIGN8(off + 4ull * c.optcap);
    c.bres = (u64 *)(ws + off);
    off += 8ull * 2 * 2 * 2 * c.alw;
    c.memo_cap = 8;
    c.memo = (u64 *)(ws + off);
    off += 8ull * 4 * c.alw * c.memo_cap;
    c.memo_err = (u32 *)(ws + off);
    off += 4ull * c.memo_cap;
    /* metrics scratch (phase C only: metric block, RECORD_BP totals, the wfa_ed wavefront, one filtered
     * sequence) sits at the TOP of the workspace and overlaps the tail of the node pool: the searches may
     * grow into it, phase C only needs the optima at the front of the pool to stay clear of it */
    const u64 csz = AVK_ALIGN8(4ull * AVK_N_GROUPS * AVK_N_FIELDS + 4ull * 32 + 4ull * c.wfs_cap + c.seqcap);
    if (off + csz + 64 > ws_bytes) return RS_OVERFLOW;
    {
        u64 t = ws_bytes - csz;
        c.cscratch_off = t;
        c.gm = (u32 *)(ws + t);
        t += 4ull * AVK_N_GROUPS * AVK_N_FIELDS;
        c.gq = (u32 *)(ws + t); /* 32 words: RECORD_BP totals per group, truth then query */
        t += 4ull * 32;
        c.wfs = (u32 *)(ws + t);
        t += 4ull * c.wfs_cap;
        c.seq_a = ws + t;
    }
    {
        /* the rest: node pool + queue (key 8 + slot 4 + free 4 bytes per possible node) */
        const u64 avail = ws_bytes - off;
        u64 qcap = avail / ((u64)c.nodeB_bytes + 16);
        if (qcap > 0x7FFFFFFFull) qcap = 0x7FFFFFFFull;
        c.qcap = (u32)qcap;
        c.qkeys = (u64 *)(ws + off);
        off += 8ull * c.qcap;
        c.qslots = (u32 *)(ws + off);
        off += 4ull * c.qcap;
        c.freelist = (u32 *)(ws + off);
        off = AVK_ALIGN16(off + 4ull * c.qcap);
        c.pool = ws + off;
        c.pool_bytes = ws_bytes > off ? ws_bytes - off : 0;
        c.pool_off = off;
    }
    if (c.qcap < 3) return RS_OVERFLOW;

    /* stage the region: reference window, variants, alleles */
    {
        /* reference window: read from the 2-bit packed genome (a quarter of the HBM bytes, one or two
         * 32-byte sectors for the usual 101-base window) unless a word of the window is flagged as
         * holding something else than A/C/G/T, in which case the raw bytes are read */
        bool packed = a.ref_2bit != (const u32 *)0;
        if (packed) {
            const u64 w0 = reg.ref_off >> 4, w1 = (reg.ref_off + (c.L ? c.L - 1 : 0)) >> 4;
            bool l_exc = false;
            for (u64 w = w0 + lane; w <= w1; w += 64) l_exc = l_exc || ((a.ref_exc[w >> 5] >> (w & 31)) & 1u);
            packed = wv_ballot(l_exc) == 0;
        }
        if (packed) {
            for (u32 b = lane * 4; b < c.L; b += 256) { /* four bases = one aligned LDS word per lane */
                const u64 p = reg.ref_off + b;
                const u64 w = p >> 4;
                const u32 sh = (u32)(p & 15) * 2;
                u64 bits = a.ref_2bit[w];
                if (sh > 24) bits |= (u64)a.ref_2bit[w + 1] << 32;
                const u32 code = (u32)(bits >> sh) & 0xFFu;
                u32 word = 0;
                for (int j = 0; j < 4; ++j) word |= ((0x54474341u >> (8 * ((code >> (2 * j)) & 3u))) & 0xFFu) << (8 * j);
                if (b + 4 <= c.L) *(u32 *)(refbuf + b) = word;
                else
                    for (u32 j = 0; b + j < c.L; ++j) refbuf[b + j] = (u8)(word >> (8 * j));
            }
        } else {
            copy_bytes(refbuf, a.ref_bytes + reg.ref_off, c.L);
        }
    }
    c.ref = refbuf;
    { /* variants, alleles, order, counts: one coalesced copy of the prepared blob */
        const u32 *src = a.blob + 2ull * wv_uni(reg.blob_off);
        const u32 nw = blob_bytes >> 2;
        for (u32 i = lane; i < nw; i += 64) blob_dst[i] = src[i];
    }
    wv_sync();

    if (TEAM && c.team) team_publish(c);
    AVK_T_MARK(c, 0)
    /* ---- phase A */
    const int nopt = phaseA<TEAM>(c);
    AVK_T_MARK(c, 1)
    if (nopt == RS_OVERFLOW) return RS_OVERFLOW;
    if (nopt < 0) return -nopt - 100;
    out.n_opt = (u32)nopt;
    if (a.mode == 1) { /* merge_solver.rs:137-143: only all_opt_haps[0].is_exact_match() is looked at, and every
                          tied optimum has the same total cost = ed1 + ed2 + the four skip distances */
        out.ed1 = c.best_cost == 0 ? 1u : 0u;
        return AVK_ST_OK;
    }

    /* keep the optima at the front of the pool: nodes [0, nopt) */
    {
        /* selection: repeatedly move optimum k to slot k (swap through a spare copy is not
         * needed: optima that sit in slots < nopt but belong elsewhere are handled by cycling) */
        for (u32 k = 0; k < (u32)nopt; ++k) {
            const u32 src = ld32u(c.optlist + k);
            if (src == k) continue;
            /* is slot k occupied by another optimum?  then swap their places via the list */
            int other = -1;
            for (u32 j = k + 1; j < (u32)nopt; ++j)
                if (ld32u(c.optlist + j) == k) other = (int)j;
            if (other >= 0) {
                /* swap contents of nodes src and k word by word */
                u32 *pa = (u32 *)node_at(c, src), *pb = (u32 *)node_at(c, k);
                wv_sync();
                for (u32 i = lane; i < (c.nodeA_bytes >> 2); i += 64) {
                    const u32 t = pa[i];
                    pa[i] = pb[i];
                    pb[i] = t;
                }
                wv_sync();
                st32(c.optlist + (u32)other, src);
            } else {
                node_copy(c, k, src);
            }
            wv_sync();
            st32(c.optlist + k, k);
            wv_sync();
        }
    }

    if (c.pool_off + (u64)nopt * c.nodeA_bytes > c.cscratch_off) return RS_OVERFLOW; /* optima would sit under the metrics scratch */

    if (a.enable_exact_shortcut && c.best_cost == 0) { /* waffle_solver.rs:171-199: returns on the first optimum */
        const HapPtr s0 = hap_ptr(c.pool + NODE_HDR, c.alw, c.wfcap, c.seqcap);
        const HapPtr s1 = hap_ptr(c.pool + NODE_HDR + c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
        const HapHdr g0 = hap_load(s0.w), g1 = hap_load(s1.w);
        winner_node = 0;
        out.ed1 = 0;
        out.ed2 = 0;
        u32 present = 0;
        const int st = exact_shortcut_metrics(a, c, v_off, s0, g0, s1, g1, present);
        out.present = present;
        return st;
    }

    /* ---- phase B for every tied optimum (waffle_solver.rs:169-261).
     * optimize_gt_alleles is a pure function of the haplotype's input alleles, and tied optima share
     * haplotypes (mirror images, independent orientation choices), so finished runs are memoised by
     * their input bit-sets.  A candidate can only win with STRICTLY fewer flips than the best so far
     * (min_by_key keeps the first minimum, :264-265), which gives every run its cutoff. */
    const u64 a_bytes = (u64)nopt * c.nodeA_bytes;
    u8 *const poolA = c.pool;
    const u32 nodeA_bytes = c.nodeA_bytes;
    const bool small_n = c.N < 500;
    u32 best_total = 0xFFFFFFFFu, best_k = 0;
    u32 n_memo = 0;
    /* The common case in one look: the first optimum costs 0 and skipped no ALT on either haplotype.  Both of its genotype
     * searches are then the trivial ones (see below), their result is the node's own allele sets with 0 flips, and total == 0
     * ends the loop at k = 0: the observed alleles ARE the expected ones, nothing is searched or copied. */
    bool obs_same = false;
    if (small_n && c.best_cost == 0) {
        const u32 *w = (const u32 *)(poolA + NODE_HDR);
        const u32 ns0 = ld32u(w + H_NSKIP), ns1 = ld32u((const u32 *)((const u8 *)w + c.hapA_bytes) + H_NSKIP);
        obs_same = ns0 == 0 && ns1 == 0;
    }
    if (TEAM && c.team && !obs_same && nopt > 0 && c.memo_cap >= 2) {
        /* The two genotype searches of the FIRST optimum are independent (one per haplotype, neither has a cutoff yet: best_total is still unset) — 27 of the 89 M
         * ticks of a 92-call window: two jobs, the owner's in its own phase-B pool, a sibling's in its scratch slice.  Their results go into the memo; the loop below
         * then finds them there and goes on exactly as if it had run them (a job that failed or outgrew its pool leaves no entry: the loop runs that search itself). */
        const HapPtr g0 = hap_ptr(poolA + NODE_HDR, c.alw, c.wfcap, c.seqcap), g1 = hap_ptr(poolA + NODE_HDR + c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
        const HapHdr gh0 = hap_load(g0.w), gh1 = hap_load(g1.w);
        const bool need0 = !(gh0.ed == 0 && gh0.nskip == 0 && small_n), need1 = !(gh1.ed == 0 && gh1.nskip == 0 && small_n);
        bool l_diff = false;
        for (u32 i = lane; i < c.alw; i += 64) l_diff = l_diff || g0.talt[i] != g1.talt[i] || g0.qalt[i] != g1.qalt[i];
        const bool differ = wv_ballot(l_diff) != 0;
        const u64 bbytes0 = c.pool_bytes > AVK_ALIGN16(a_bytes) ? c.pool_bytes - AVK_ALIGN16(a_bytes) : 0;
        u64 cap0 = bbytes0 / c.nodeB_bytes;
        if (cap0 > c.qcap) cap0 = c.qcap;
        if (need0 && need1 && differ && cap0 >= 3) {
            wv_sync();
            for (u32 i = lane; i < c.alw; i += 64) {
                c.memo[i] = g0.talt[i], c.memo[c.alw + i] = g0.qalt[i];
                c.memo[4ull * c.alw + i] = g1.talt[i], c.memo[5ull * c.alw + i] = g1.qalt[i];
            }
            wv_sync();
            if (lane == 0) {
                TeamBox *tb = c.team;
                tb->own_poolB = (u64)(poolA + AVK_ALIGN16(a_bytes));
                for (u32 j = 0; j < 2; ++j) {
                    TeamJob *jb = tb->job + j;
                    jb->kind = TJ_GT, jb->flags = 0, jb->depth = jb->allele = 0;
                    jb->x0 = (u32)cap0;
                    jb->x1 = (j == 0 || c.team_waves <= 1) ? 1u : 0u; /* (the owner alone: both in its own pool, one after the other) */
                    jb->src = (u64)(c.memo + (u64)j * 4 * c.alw), jb->dst = 0;
                }
            }
            team_run(c, 2);
            const int e0 = (int)ld32u((const u32 *)&c.team->job[0].res[0]), e1 = (int)ld32u((const u32 *)&c.team->job[1].res[0]);
            if (e0 >= 0) {
                st32(c.memo_err + 0, (u32)e0);
                n_memo = 1;
                if (e1 >= 0) {
                    st32(c.memo_err + 1, (u32)e1);
                    n_memo = 2;
                }
                wv_sync();
            }
        }
    }
    for (u32 k = 0; k < (obs_same ? 0u : (u32)nopt); ++k) {
        u32 errs[2] = {0, 0};
        bool cut = false;
        for (int hh = 0; hh < 2 && !cut; ++hh) {
            const HapPtr ap = hap_ptr(poolA + (u64)k * nodeA_bytes + NODE_HDR + (u64)hh * c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
            u64 *res = c.bres + (u64)(0 * 2 + hh) * 2 * c.alw; /* candidate slot 0 = current */
            const HapHdr ah = hap_load(ap.w);
            if (ah.ed == 0 && ah.nskip == 0 && small_n) {
                /* The phasing search already proved truth == query on this haplotype with every ALT
                 * incorporated.  The exact-match search then has exactly one zero-error path (keep every
                 * allele); it is always the top of the queue, cannot be pruned before it finalises
                 * (fewer than 500 expansions, :309-311) and wins with num_errors = 0 and the input
                 * alleles (exact_gt_optimizer.rs:169-192).  Nothing to search. */
                wv_sync();
                for (u32 i = lane; i < c.alw; i += 64) {
                    res[i] = ap.talt[i];
                    res[c.alw + i] = ap.qalt[i];
                }
                wv_sync();
                errs[hh] = 0;
                continue;
            }
            /* memo lookup: entry m = [talt | qalt | res_t | res_q] (4*alw words) + errors */
            int hit = -1;
            for (u32 m = 0; m < n_memo && hit < 0; ++m) {
                const u64 *e = c.memo + (u64)m * 4 * c.alw;
                bool l_diff = false;
                for (u32 i = lane; i < c.alw; i += 64) l_diff = l_diff || e[i] != ap.talt[i] || e[c.alw + i] != ap.qalt[i];
                if (wv_ballot(l_diff) == 0) hit = (int)m;
            }
            if (hit >= 0) {
                const u64 *e = c.memo + (u64)hit * 4 * c.alw;
                wv_sync();
                for (u32 i = lane; i < 2 * c.alw; i += 64) res[i] = e[2 * c.alw + i];
                wv_sync();
                errs[hh] = ld32u(c.memo_err + hit);
                continue;
            }
            u32 cutoff = 0xFFFFFFFFu;
            if (small_n && best_total != 0xFFFFFFFFu) cutoff = best_total > errs[0] ? best_total - (hh ? errs[0] : 0u) : 0u;
            if (cutoff == 0) {
                cut = true;
                break;
            }
            c.node_bytes = c.nodeB_bytes;
            c.pool_base = poolA + AVK_ALIGN16(a_bytes);
            const u64 bbytes = c.pool_bytes > AVK_ALIGN16(a_bytes) ? c.pool_bytes - AVK_ALIGN16(a_bytes) : 0;
            u64 cap = bbytes / c.nodeB_bytes;
            if (cap > c.qcap) cap = c.qcap;
            c.pool_cap = (u32)cap;
            const int e = phaseB(c, ap.talt, ap.qalt, res, cutoff);
            if (e == RS_OVERFLOW) return RS_OVERFLOW;
            if (e == RS_CUT) {
                cut = true;
                break;
            }
            if (e < 0) return -e - 100;
            errs[hh] = (u32)e;
            if (n_memo < c.memo_cap) {
                u64 *me = c.memo + (u64)n_memo * 4 * c.alw;
                wv_sync();
                for (u32 i = lane; i < c.alw; i += 64) {
                    me[i] = ap.talt[i];
                    me[c.alw + i] = ap.qalt[i];
                }
                for (u32 i = lane; i < 2 * c.alw; i += 64) me[2 * c.alw + i] = res[i];
                st32(c.memo_err + n_memo, (u32)e);
                wv_sync();
                n_memo += 1;
            }
        }
        if (cut) continue; /* this optimum needs at least as many flips as the best one */
        const u32 total = errs[0] + errs[1];
        if (total < best_total) { /* min_by_key keeps the FIRST minimum, :264-265 */
            best_total = total;
            best_k = k;
            wv_sync();
            for (u32 i = lane; i < 4 * c.alw; i += 64) c.bres[4 * c.alw + i] = c.bres[i]; /* candidate slot 1 = best */
            wv_sync();
            if (total == 0) break; /* no later optimum can be strictly better */
        }
    }
    winner_node = best_k;
    const u64 *obs = c.bres + 4 * c.alw; /* [hap][side][alw] */
    AVK_T_MARK(c, 2)

    /* ---- phase C */
    u8 *wn = poolA + (u64)best_k * nodeA_bytes;
    const HapPtr w0 = hap_ptr(wn + NODE_HDR, c.alw, c.wfcap, c.seqcap);
    const HapPtr w1 = hap_ptr(wn + NODE_HDR + c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
    const HapHdr h0 = hap_load(w0.w), h1 = hap_load(w1.w);
    out.ed1 = h0.ed;
    out.ed2 = h1.ed;

    if (a.group_metrics) zero_words(c.gm, AVK_N_GROUPS * AVK_N_FIELDS); /* the whole block goes out */
    else gm_zero(c.gm, 1u | ((wv_uni(reg.pre_status) >> 16) << 1));
    zero_words(c.gq, 32);
    wv_sync();
    /* compare_expected_observed for truth and query (:296-327) + per-variant outputs */
    u32 l_bad = 0;
    for (u32 k = lane; k < c.N; k += 64) {
        const bool is_truth = k < c.T;
        const u32 sub = is_truth ? k : k - c.T;
        const u64 *e0 = is_truth ? w0.talt : w0.qalt, *e1 = is_truth ? w1.talt : w1.qalt;
        const u64 *o0 = obs_same ? e0 : obs + (is_truth ? 0 : c.alw), *o1 = obs_same ? e1 : obs + 2 * c.alw + (is_truth ? 0 : c.alw);
        const u32 b0 = (u32)((e0[sub >> 6] >> (sub & 63)) & 1), b1 = (u32)((e1[sub >> 6] >> (sub & 63)) & 1);
        const u32 exp = b0 + b1;
        const u32 ob = (u32)((o0[sub >> 6] >> (sub & 63)) & 1) + (u32)((o1[sub >> 6] >> (sub & 63)) & 1);
        const LVar v = c.vars[k]; /* per-lane variant: lanes work on different variants here */
        const u32 vtype = v.type_zyg & 0xFF;
        if (exp == 0) l_bad = AVK_ST_VARIANT_METRICS;
        else if (exp < ob) l_bad = AVK_ST_TRUTH_FP;
        else gm_add(c.gm, !is_truth, vtype, v.alt_ed, exp, ob);
        /* VariantMetrics (variant_metrics.rs:43-101); query entries are toggled */
        const u32 gv = v_off + k;
        u32 cls = exp == ob ? AVK_CLASS_TP : AVK_CLASS_FN;
        u32 ea = exp, oa = ob;
        if (!is_truth) {
            if (cls == AVK_CLASS_FN) cls = AVK_CLASS_FP;
            ea = ob;
            oa = exp;
        }
        const u32 rz = b0 && b1 ? AVK_ZYG_HOM_ALT : (b0 ? AVK_ZYG_PHASED_HET10 : AVK_ZYG_PHASED_HET01);
        a.var_out[gv] = ea | (oa << 8) | (cls << 16) | (rz << 24);
    }
    const u32 bad = wv_max_u32(l_bad);
    if (bad) return (int)bad;
    wv_sync();
    /* add_basepair_stats (:335-449).  The optimizer's own sequences are the regenerated ones
     * (asserted equal at :364-367), ed(truth,query) is the node's finalized DWFA distance. */
    const int SUP[8] = {AVK_VT_SNV, AVK_VT_INSERTION, AVK_VT_DELETION, AVK_VT_INDEL, AVK_VT_TR_CONTRACTION, AVK_VT_TR_EXPANSION, AVK_VT_SV_DELETION, AVK_VT_SV_INSERTION};
    u32 present = wv_uni(reg.pre_status) >> 16; /* types seen by compare_expected_observed, plus the 8 filtered types' map entries */
    for (int s = 0; s < 8; ++s) present |= 1u << SUP[s];
    AVK_T_MARK(c, 3)
    /* which of the 8 filtered types occur at all: one LDS read by 8 lanes instead of 8 dependent round trips per haplotype */
    const u32 type_mask = (u32)wv_ballot(lane < 8 && c.counts[lane < 8 ? lane : 0] != 0) & 0xFFu;
    const bool team_metrics = TEAM && c.team && 4ull * c.wfs_cap + c.seqcap + 64 <= c.team_scratch_bytes;
    if (team_metrics) {
        /* the same sums, their alignments dealt out over the team: first the (at most four) distances to the reference window, then one job per
         * (haplotype, call type, side) that needs a filtered sequence — each makes its sequence in its wave's own scratch and aligns it twice */
        const HapPtr *wps[2] = {&w0, &w1};
        const HapHdr *hds[2] = {&h0, &h1};
        bool t_alt[2], q_alt[2];
        int jt[2] = {-1, -1}, jq[2] = {-1, -1};
        u32 nj = 0;
        for (int hh = 0; hh < 2; ++hh) {
            bool l_t = false, l_q = false;
            for (u32 i = lane; i < c.alw; i += 64) {
                l_t = l_t || wps[hh]->talt[i] != 0;
                l_q = l_q || wps[hh]->qalt[i] != 0;
            }
            t_alt[hh] = wv_ballot(l_t) != 0, q_alt[hh] = wv_ballot(l_q) != 0;
            if (t_alt[hh]) jt[hh] = (int)nj++;
            if (q_alt[hh] && !(hds[hh]->ed == 0 && t_alt[hh])) jq[hh] = (int)nj++;
        }
        wv_sync();
        if (lane == 0)
            for (int hh = 0; hh < 2; ++hh) {
                if (jt[hh] >= 0) {
                    TeamJob *jb = c.team->job + jt[hh];
                    jb->kind = TJ_ED, jb->flags = 0, jb->depth = jb->allele = 0, jb->src = (u64)c.ref, jb->x0 = c.L, jb->dst = (u64)wps[hh]->tseq, jb->x1 = hds[hh]->t_len;
                }
                if (jq[hh] >= 0) {
                    TeamJob *jb = c.team->job + jq[hh];
                    jb->kind = TJ_ED, jb->flags = 0, jb->depth = jb->allele = 0, jb->src = (u64)c.ref, jb->x0 = c.L, jb->dst = (u64)wps[hh]->qseq, jb->x1 = hds[hh]->q_len;
                }
            }
        if (nj) team_run(c, nj);
        u32 X[2], Y[2], TP[2];
        for (int hh = 0; hh < 2; ++hh) {
            int ert = 0, erq = 0;
            if (jt[hh] >= 0) {
                ert = (int)wv_uni((u32)c.team->job[jt[hh]].res[0]);
                if (ert < 0) return RS_OVERFLOW;
            }
            if (q_alt[hh]) {
                if (hds[hh]->ed == 0 && t_alt[hh]) erq = ert;
                else {
                    erq = (int)wv_uni((u32)c.team->job[jq[hh]].res[0]);
                    if (erq < 0) return RS_OVERFLOW;
                }
            } else if (hds[hh]->ed == 0) erq = ert;
            X[hh] = 2u * (u32)ert, Y[hh] = 2u * (u32)erq;
            TP[hh] = (X[hh] + Y[hh] - 2u * hds[hh]->ed) / 2;
            u32 add[4] = {TP[hh], X[hh] - TP[hh] + 2 * hds[hh]->t_skip, TP[hh], Y[hh] - TP[hh] + 2 * hds[hh]->q_skip}; /* + skip metrics :378-381 */
            wv_sync();
            if (lane < 4) c.gm[AVK_F_BP_TRUTH_TP + lane] += add[lane];
            wv_sync();
        }
        /* the filtered alignments: job index of (hh, s, side) or -1 */
        int fj[2][8][2];
        nj = 0;
        for (int hh = 0; hh < 2; ++hh)
            for (int s = 0; s < 8; ++s) {
                fj[hh][s][0] = fj[hh][s][1] = -1;
                if (!((type_mask >> s) & 1u)) continue;
                const u32 tq = ld32u(c.counts + s), tcount_s = tq & 0xFFFFu, qcount_s = tq >> 16;
                if (qcount_s && qcount_s != c.Q) fj[hh][s][1] = (int)nj++;
                if (tcount_s && tcount_s != c.T) fj[hh][s][0] = (int)nj++;
            }
        wv_sync();
        if (lane == 0)
            for (int hh = 0; hh < 2; ++hh)
                for (int s = 0; s < 8; ++s)
                    for (int side = 0; side < 2; ++side)
                        if (fj[hh][s][side] >= 0) {
                            TeamJob *jb = c.team->job + fj[hh][s][side];
                            jb->kind = TJ_FILT, jb->flags = 0, jb->depth = jb->allele = 0, jb->x0 = (u32)side, jb->x1 = (u32)s, jb->src = (u64)wps[hh]->w, jb->dst = 0;
                        }
        if (nj) team_run(c, nj);
        for (int hh = 0; hh < 2; ++hh)
            for (u32 left = type_mask; left; left &= left - 1) {
                const int s = avk_ctz64(left);
                const u32 tq = ld32u(c.counts + s), tcount_s = tq & 0xFFFFu, qcount_s = tq >> 16;
                const u32 tp = TP[hh];
                u32 q_tp = 0, q_fp = 0, t_tp = 0, t_fn = 0;
                if (qcount_s) {
                    if (qcount_s == c.Q) {
                        q_tp = tp;
                        q_fp = Y[hh] - tp + 2 * hds[hh]->q_skip;
                    } else {
                        const TeamJob *jb = c.team->job + fj[hh][s][1];
                        const int y2 = (int)wv_uni((u32)jb->res[0]), z2 = (int)wv_uni((u32)jb->res[1]);
                        if (y2 < 0 || z2 < 0) return RS_OVERFLOW;
                        const u32 Y2 = 2u * (u32)y2, Z2 = 2u * (u32)z2;
                        const u32 tp2 = (X[hh] + Y2 - Z2) / 2;
                        q_tp = tp2;
                        q_fp = Y2 - tp2 + 2 * wv_uni((u32)jb->res[2]);
                    }
                }
                if (tcount_s) {
                    if (tcount_s == c.T) {
                        t_tp = tp;
                        t_fn = X[hh] - tp + 2 * hds[hh]->t_skip;
                    } else {
                        const TeamJob *jb = c.team->job + fj[hh][s][0];
                        const int x2 = (int)wv_uni((u32)jb->res[0]), z2 = (int)wv_uni((u32)jb->res[1]);
                        if (x2 < 0 || z2 < 0) return RS_OVERFLOW;
                        const u32 X2 = 2u * (u32)x2, Z2 = 2u * (u32)z2;
                        const u32 tp2 = (X2 + Y[hh] - Z2) / 2;
                        t_tp = tp2;
                        t_fn = X2 - tp2 + 2 * wv_uni((u32)jb->res[2]);
                    }
                }
                wv_sync();
                if (lane == 0) {
                    u32 *g = c.gm + (1 + SUP[s]) * AVK_N_FIELDS;
                    g[AVK_F_BP_TRUTH_TP] += t_tp;
                    g[AVK_F_BP_TRUTH_FN] += t_fn;
                    g[AVK_F_BP_QUERY_TP] += q_tp;
                    g[AVK_F_BP_QUERY_FP] += q_fp;
                }
                wv_sync();
            }
    }
    for (int hh = 0; hh < (team_metrics ? 0 : 2); ++hh) {
        const HapPtr &wp = hh == 0 ? w0 : w1;
        const HapHdr &hd = hh == 0 ? h0 : h1;
        /* a haplotype that carries no ALT allele IS the reference window: distance 0 without aligning */
        bool l_t = false, l_q = false;
        for (u32 i = lane; i < c.alw; i += 64) {
            l_t = l_t || wp.talt[i] != 0;
            l_q = l_q || wp.qalt[i] != 0;
        }
        const bool t_has_alt = wv_ballot(l_t) != 0, q_has_alt = wv_ballot(l_q) != 0;
        int ert = 0, erq = 0;
        if (t_has_alt) {
            ert = wfa_ed(c, c.ref, c.L, wp.tseq, hd.t_len);
            if (ert < 0) return RS_OVERFLOW;
        }
        if (q_has_alt) {
            if (hd.ed == 0 && t_has_alt) erq = ert; /* distance 0 after finalize: the two sequences are identical */
            else {
                erq = wfa_ed(c, c.ref, c.L, wp.qseq, hd.q_len);
                if (erq < 0) return RS_OVERFLOW;
            }
        } else if (hd.ed == 0) erq = ert;
        const u32 X = 2u * (u32)ert, Y = 2u * (u32)erq, Z = 2u * hd.ed;
        const u32 tp = (X + Y - Z) / 2;
        u32 add[4] = {tp, X - tp + 2 * hd.t_skip, tp, Y - tp + 2 * hd.q_skip}; /* + skip metrics :378-381 */
        wv_sync();
        if (lane < 4) c.gm[AVK_F_BP_TRUTH_TP + lane] += add[lane];
        wv_sync();
        for (u32 left = type_mask; left; left &= left - 1) { /* absent types contribute (0,0,0,0); their map entries exist via `present` */
            const int s = avk_ctz64(left);
            const u32 tq = ld32u(c.counts + s), tcount_s = tq & 0xFFFFu, qcount_s = tq >> 16;
            u32 q_tp = 0, q_fp = 0, t_tp = 0, t_fn = 0;
            if (qcount_s) {
                if (qcount_s == c.Q) { /* filtered query == the full query haplotype */
                    q_tp = tp;
                    q_fp = Y - tp + 2 * hd.q_skip;
                } else {
                    u32 failed;
                    const u32 fl = gen_filtered(c, c.T, c.Q, wp.qalt, (u32)SUP[s], c.seq_a, failed);
                    const int y2 = wfa_ed(c, c.ref, c.L, c.seq_a, fl);
                    if (y2 < 0) return RS_OVERFLOW;
                    const int z2 = wfa_ed(c, wp.tseq, hd.t_len, c.seq_a, fl);
                    if (z2 < 0) return RS_OVERFLOW;
                    const u32 Y2 = 2u * (u32)y2, Z2 = 2u * (u32)z2;
                    const u32 tp2 = (X + Y2 - Z2) / 2;
                    q_tp = tp2;
                    q_fp = Y2 - tp2 + 2 * failed;
                }
            }
            if (tcount_s) {
                if (tcount_s == c.T) {
                    t_tp = tp;
                    t_fn = X - tp + 2 * hd.t_skip;
                } else {
                    u32 failed;
                    const u32 fl = gen_filtered(c, 0, c.T, wp.talt, (u32)SUP[s], c.seq_a, failed);
                    const int x2 = wfa_ed(c, c.ref, c.L, c.seq_a, fl);
                    if (x2 < 0) return RS_OVERFLOW;
                    const int z2 = wfa_ed(c, c.seq_a, fl, wp.qseq, hd.q_len);
                    if (z2 < 0) return RS_OVERFLOW;
                    const u32 X2 = 2u * (u32)x2, Z2 = 2u * (u32)z2;
                    const u32 tp2 = (X2 + Y - Z2) / 2;
                    t_tp = tp2;
                    t_fn = X2 - tp2 + 2 * failed;
                }
            }
            wv_sync();
            if (lane == 0) {
                u32 *g = c.gm + (1 + SUP[s]) * AVK_N_FIELDS;
                g[AVK_F_BP_TRUTH_TP] += t_tp;
                g[AVK_F_BP_TRUTH_FN] += t_fn;
                g[AVK_F_BP_QUERY_TP] += q_tp;
                g[AVK_F_BP_QUERY_FP] += q_fp;
            }
            wv_sync();
        }
    }

    AVK_T_MARK(c, 4)
    /* add_record_basepair_stats (:455-522): totals from the INPUT zygosities and raw allele space */
    wv_sync();
    {
        u32 *tot = c.gq; /* tot[0..12] truth totals per group, tot[16..28] query totals (zeroed above) */
        for (u32 k = lane; k < c.N; k += 64) {
            const LVar v = c.vars[k];
            const u32 z = (v.type_zyg >> 8) & 0xFF;
            const u32 cntz = z == AVK_ZYG_HOM_ALT ? 2u : ((z == AVK_ZYG_UNPHASED_HET || z == AVK_ZYG_PHASED_HET01 || z == AVK_ZYG_PHASED_HET10) ? 1u : 0u);
            const u32 val = cntz * v.raw_space;
            const u32 basei = k < c.T ? 0 : 16;
            avk_atomic_add_u32(tot + basei, val);
            avk_atomic_add_u32(tot + basei + 1 + (v.type_zyg & 0xFF), val);
        }
        wv_sync();
        u32 l_err = 0;
        for (u32 g = lane; g < AVK_N_GROUPS; g += 64) {
            const bool on = g == 0 || ((present >> (g - 1)) & 1);
            if (on) {
                u32 *d = c.gm + g * AVK_N_FIELDS;
                const u32 tfn = d[AVK_F_BP_TRUTH_FN], qfp = d[AVK_F_BP_QUERY_FP];
                const u32 ttp = 2 * tot[g] - tfn, qtp = 2 * tot[16 + g] - qfp;
                if (g == 0 && (ttp < d[AVK_F_BP_TRUTH_TP] || qtp < d[AVK_F_BP_QUERY_TP])) l_err = AVK_ST_RECORD_BP;
                d[AVK_F_RBP_TRUTH_TP] += ttp;
                d[AVK_F_RBP_TRUTH_FN] += tfn;
                d[AVK_F_RBP_QUERY_TP] += qtp;
                d[AVK_F_RBP_QUERY_FP] += qfp;
            }
        }
        const u32 e = wv_max_u32(l_err);
        if (e) return (int)e;
    }
    wv_sync();
    AVK_T_MARK(c, 5)
    out.present = present;
    return AVK_ST_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* persistent wave: pulls regions, walks the tiers, writes results, keeps a private tally      */
/* ------------------------------------------------------------------------------------------ */
AVK_DEV void write_region_record(const AvkKernelArgs &a, u32 r, int status, u32 ed1, u32 ed2, u32 nopt, u32 present) {
    const u32 lane = (u32)wv_lane();
    if (lane < 4) {
        const u32 w = lane == 0 ? (u32)status : lane == 1 ? ed1 : lane == 2 ? ed2 : (nopt | (present << 16));
        a.region_out[4 * (u64)r + lane] = w; /* 4 lanes, one 16-byte segment */
    }
}

/* rec = index of the region's record (work order); the outputs sit at the region's place in the caller's batch */
AVK_DEV void write_failed_region(const AvkKernelArgs &a, u32 rec, int status) {
    const AvkDevRegion reg = a.regions[rec];
    const u32 lane = (u32)wv_lane();
    const u32 r = wv_uni(reg.orig);
    write_region_record(a, r, status, 0, 0, 0, 0);
    const u32 n = reg.t_cnt + reg.q_cnt;
    for (u32 k = lane; k < n; k += 64) a.var_out[reg.v_off + k] = 0;
    if (a.group_metrics) zero_words(a.group_metrics + (u64)r * AVK_N_GROUPS * AVK_N_FIELDS, AVK_N_GROUPS * AVK_N_FIELDS);
    if (a.bp_out) zero_words(a.bp_out + 4 * (u64)a.bp_off[r], 4 * (a.bp_off[r + 1] - a.bp_off[r]));
    if (a.seq_bytes && a.seq_len && lane < 5) a.seq_len[5 * (u64)r + lane] = 0;
}

/* PASS_LDS: this launch solves regions in the wave's LDS slice; otherwise in its HBM slice
 * (a.pass_tier says which tier's capacities apply).  A region that does not fit is appended to the
 * overflow list for the next launch; after the last tier it fails with AVK_ST_CAPACITY. */
/* ---- in-workgroup escalation (LDS launches): control words at the end of the workgroup's LDS.
 * ctl[0] lock = ticket of the wave that owns the whole LDS (0 = nobody), ctl[1] ticket counter,
 * ctl[2 + w] state of wave w: 0 running, a ticket = parked for that ticket, 0xFFFFFFFF gone.
 * A wave parks only between regions (nothing in its slice is live then); the owner waits until every sibling
 * acknowledged ITS ticket, so a sibling that is just leaving an earlier park cannot slip back into its slice. */
AVK_DEV u32 wg_word(u32 *p) {
    u32 v = 0;
    if (wv_lane() == 0) v = avk_wg_load(p);
    return wv_uni(wv_shfl(v, 0));
}
AVK_DEV void wg_park(u32 *ctl, u32 w, u32 ticket) {
    for (;;) {
        if (wv_lane() == 0) avk_wg_store(ctl + 2 + w, ticket);
        u32 cur;
        do {
            avk_sleep();
            cur = wg_word(ctl);
        } while (cur == ticket);
        if (cur == 0) {
            if (wv_lane() == 0) avk_wg_store(ctl + 2 + w, 0u);
            return; /* the caller looks at the lock again before it touches its slice */
        }
        ticket = cur;
    }
}
AVK_DEV void wg_acquire(u32 *ctl, u32 w, u32 n_wg_waves) {
    u32 ticket;
    for (;;) {
        u32 t = 0, old = 0;
        if (wv_lane() == 0) {
            t = avk_wg_add(ctl + 1, 1u) + 1u;
            old = avk_wg_cas(ctl, 0u, t);
        }
        ticket = wv_uni(wv_shfl(t, 0));
        old = wv_uni(wv_shfl(old, 0));
        if (old == 0) break;
        wg_park(ctl, w, old); /* a sibling owns the LDS: it needs this wave's slice too */
    }
    for (u32 o = 0; o < n_wg_waves; ++o) {
        if (o == w) continue;
        for (;;) {
            const u32 s = wg_word(ctl + 2 + o);
            if (s == ticket || s == 0xFFFFFFFFu) break;
            avk_sleep();
        }
    }
}

/* With esc_bytes set (bulk launch) the workgroups have exactly 4 waves: wave w of a workgroup has wave_id % 4 == w and
 * its slice starts w slices into the workgroup's LDS. */
/* LAZY: a launch for regions the lanes handed back (device-packed batches write those regions' records on demand, below) — a separate instantiation, so that
 * the launches that solve a genome's bulk keep their register allocation */
template <bool PASS_LDS, bool LAZY = false, bool TEAM = false> AVK_DEV void region_worker(const AvkKernelArgs &a, u32 wave_id, u8 *lds_slice, TeamBox *team = (TeamBox *)0) {
    const bool wgt = PASS_LDS && a.esc_bytes != 0; /* this workgroup has a tail: shared tally, and the lock of the escalation */
    const bool esc = wgt && a.esc_enabled != 0;
    const u32 wave_in_wg = wave_id & 3u;
    u8 *const wg_lds = lds_slice - (PASS_LDS ? (u32)wave_in_wg * (u32)a.tier[a.pass_tier].ws_bytes : 0u); /* meaningful when esc */
    const u32 n_wg_waves = 4;
    const u32 lane = (u32)wv_lane();
    u32 n_ok = 0, n_err = 0, n_cap = 0, n_big = 0; /* n_big: finished (either way) in a tier-3 slice */
    const u32 tier = a.pass_tier;
    u8 *ws = PASS_LDS ? lds_slice : a.hbm_ws + (u64)wave_id * a.tier[tier].ws_bytes;
    const u64 ws_bytes = a.tier[tier].ws_bytes;
    const u32 ed_cap = a.tier[tier].ed_cap;
    const u32 n_work = a.n_work_dev ? wv_uni(*a.n_work_dev) : a.n_work;

    /* Work distribution.  Returning atomics on one cache line saturate near 88 claims/us on this chip
     * (MI355X_MICROARCH.md "dequeue"), far below what the solver needs, so claims are rationed:
     *   - the first static_pct % (AVK_STATIC_PCT) of the work list is dealt statically, item k to wave k mod n_waves
     *     (no atomics, neighbouring waves read neighbouring records);
     *   - the rest is claimed dynamically in chunks of AVK_CLAIM from 8 shard counters that sit in
     *     separate 128-byte lines; a wave starts on the shard of its workgroup's XCD and peeks at a
     *     counter with a plain load before spending an atomic on it, so drained shards cost nothing.
     * The dynamic tail evens out waves that drew expensive regions. */
    const u32 n_static = a.n_waves ? (u32)(((u64)n_work * a.static_pct / 100) / a.n_waves) * a.n_waves : 0;
    const u32 n_dyn = n_work - n_static;
    const u32 n_shards = a.n_shards;
    const u32 shard_len = (n_dyn + n_shards - 1) / n_shards;
    const u32 home = (wave_id >> 2) % n_shards;
    u32 static_next = wave_id < a.n_waves ? wave_id : 0xFFFFFFFFu; /* waves beyond n_waves (placed late) only claim */
    u32 shard_i = 0, claim_base = 0, claim_left = 0;
    u32 *const wg_ctl = (u32 *)(wg_lds + a.esc_bytes); /* used when wgt: [0] lock, [1] ticket counter, [2..5] wave states, [6] waves gone */
    u32 *const wg_tally = wg_ctl + 16;
    for (;;) {
        if (esc) { /* a sibling wants the whole LDS: stay out of the slice until it is done */
            for (u32 t = wg_word(wg_ctl); t != 0; t = wg_word(wg_ctl)) wg_park(wg_ctl, wave_in_wg, t);
        }
        u32 idx = 0;
        bool have = true;
        if (static_next < n_static) {
            idx = static_next;
            static_next += a.n_waves;
        } else {
            if (claim_left == 0) {
                bool got = false;
                while (!got && shard_i < n_shards) {
                    u32 sh = home + shard_i;
                    if (sh >= n_shards) sh -= n_shards;
                    const u32 lo = sh * shard_len < n_dyn ? sh * shard_len : n_dyn;
                    const u32 hi = lo + shard_len < n_dyn ? lo + shard_len : n_dyn;
                    u32 *ctr = a.work_counter + 32u * sh;
                    u32 b = 0xFFFFFFFFu;
                    if (lane == 0) {
                        const u32 seen = *(volatile u32 *)ctr;
                        if (lo + seen < hi) b = avk_atomic_add_u32_global(ctr, a.claim);
                    }
                    b = wv_uni(wv_shfl(b, 0));
                    if (b != 0xFFFFFFFFu && lo + b < hi) {
                        claim_base = n_static + lo + b;
                        claim_left = hi - (lo + b) < a.claim ? hi - (lo + b) : a.claim;
                        got = true;
                    } else {
                        shard_i += 1;
                    }
                }
                have = got;
            }
            if (have) {
                idx = claim_base;
                claim_base += 1;
                claim_left -= 1;
            }
        }
        u32 r; /* index of the RECORD (work order) */
        if (have) {
            r = a.work_list ? wv_uni(a.work_list[idx]) : a.work_base + idx;
        } else { /* the launch's own list is done: the shared list (AvkKernelArgs::extra_counter), one record per ticket */
            if (a.extra_n == 0) break;
            u32 b = 0xFFFFFFFFu;
            if (lane == 0) {
                const u32 seen = avk_ld_agent_u32(a.extra_counter);
                if (seen < a.extra_n) b = avk_atomic_add_u32_global(a.extra_counter, 1u);
            }
            b = wv_uni(wv_shfl(b, 0));
            if (b == 0xFFFFFFFFu || b >= a.extra_n) break;
            r = a.extra_base + b;
        }
        if (LAZY && a.lazy_dp && r >= a.lazy_from) { /* a lane-class region of a device-packed batch: its record and blob are written now, by this wave */
            if (lane == 0) dp::dp_region_record(*(const dp::DpArgs *)a.lazy_dp, r);
            wv_sync();
        }
        const AvkDevRegion reg = a.regions[r];
        if (a.only_not_wide && avk_wide_static_ok(wv_uni(reg.len), wv_uni(reg.grow), wv_uni(reg.ed_bound), wv_uni(reg.t_cnt), wv_uni(reg.q_cnt), wv_uni(reg.pre_status))) continue;
        const u32 orig = wv_uni(reg.orig); /* where the caller's batch has this region: outputs go there */
        const u32 pre = wv_uni(reg.pre_status) & 0xFFFFu;
        if (pre) {
            write_failed_region(a, r, pre == AVK_PRE_SKIP_OK ? 0 : (int)pre);
            n_err += pre == AVK_PRE_SKIP_OK ? 0u : 1u;
            n_ok += pre == AVK_PRE_SKIP_OK ? 1u : 0u;
            continue;
        }
        Ctx c;
        c.team = TEAM ? team : (TeamBox *)0;
        c.team_gen = TEAM && team ? (wv_uni(avk_wg_load(&team->next)) >> 8) : 0u; /* (generations go on from where the region before left them) */
        c.team_scratch_bytes = TEAM ? ws_bytes : 0;
        c.team_mode = a.team;
        c.team_waves = TEAM && team && a.team == 1 ? 4u : 1u;
        c.team_dbg = a.work_counter ? a.work_counter + 17 : (u32 *)0; /* (the team launch's claim counter is word 1256 of the batch's counters: these are 1273..1279) */
#ifdef AVK_PHASE_TIMING
        for (int k = 0; k < 16; ++k) c.tphase[k] = 0;
        const u64 t_region0 = avk_clock();
#endif
        RegionOut out;
        out.ed1 = out.ed2 = out.n_opt = out.present = 0;
        u32 winner = 0;
        /* one call site: a wave of the HBM launch that finds its tier-2 slice too small claims one of the shared
         * tier-3 slices and goes round again */
        u8 *cur_ws = ws;
        u64 cur_bytes = ws_bytes;
        u32 cur_cap = ed_cap, cur_tier = tier, slot = 0xFFFFFFFFu;
        int st;
        for (;;) {
            st = solve_region_tier<TEAM>(a, r, cur_ws, cur_bytes, cur_cap, c, out, winner);
            if (st != RS_OVERFLOW || slot != 0xFFFFFFFFu) break;
            if (PASS_LDS) {
                if (!esc) break;
                wg_acquire(wg_ctl, wave_in_wg, n_wg_waves);
                slot = 0; /* owns the workgroup's LDS */
                cur_ws = wg_lds;
                cur_bytes = a.esc_bytes;
                cur_cap = a.tier[1].ws_bytes ? a.tier[1].ed_cap : ed_cap;
                cur_tier = 1;
                continue;
            }
            if (a.big_slots == 0) break;
            u32 got = 0xFFFFFFFFu;
            if (lane == 0) {
                for (u32 probe = wave_id % a.big_slots;; probe = probe + 1 < a.big_slots ? probe + 1 : 0) { /* holders never wait: this ends */
                    if (avk_ld_agent_u32(a.big_busy + probe) == 0 && avk_atomic_cas_u32_global(a.big_busy + probe, 0u, 1u) == 0u) {
                        got = probe;
                        break;
                    }
                    avk_sleep();
                }
            }
            slot = wv_uni(wv_shfl(got, 0));
            avk_acquire_agent(); /* the slice was last written through another XCD's L2 */
            cur_ws = a.big_ws + (u64)slot * a.tier[3].ws_bytes;
            cur_bytes = a.tier[3].ws_bytes;
            cur_cap = a.tier[3].ed_cap;
            cur_tier = 3;
        }
#define AVK_RELEASE_SLOT()                                        \
    if (slot != 0xFFFFFFFFu) {                                    \
        if (PASS_LDS) {                                           \
            wv_sync();                                            \
            if (lane == 0) avk_wg_store(wg_ctl, 0u);              \
        } else {                                                  \
            avk_release_agent();                                  \
            if (lane == 0) avk_st_agent_u32(a.big_busy + slot, 0u); \
        }                                                         \
    }
        if (st == RS_OVERFLOW) {
            if (a.overflow_list) { /* hand over to the next tier's launch */
                if (lane == 0) {
                    const u32 slot_o = avk_atomic_add_u32_global(a.overflow_count, 1);
                    a.overflow_list[slot_o] = r;
                }
                AVK_RELEASE_SLOT()
                continue;
            }
            st = AVK_ST_CAPACITY;
            n_cap += 1;
        }
        if (cur_tier != tier && st != AVK_ST_CAPACITY) n_big += 1;
        if (st != AVK_ST_OK) {
            write_failed_region(a, r, st);
            n_err += 1;
            AVK_RELEASE_SLOT()
            continue;
        }
        /* results of an Ok region */
        write_region_record(a, orig, 0, out.ed1, out.ed2, out.n_opt, out.present);
        n_ok += 1;
        if (a.mode == 1) {
            AVK_RELEASE_SLOT()
            continue;
        }
        if (a.group_metrics) copy_words(a.group_metrics + (u64)orig * AVK_N_GROUPS * AVK_N_FIELDS, c.gm, AVK_N_GROUPS * AVK_N_FIELDS);
        if (a.bp_out) { /* the BASEPAIR counters of the groups that can hold anything: the joint one, then the call types of the region in type order */
            u32 *dst = a.bp_out + 4 * (u64)wv_uni(a.bp_off[orig]);
            for (u32 left = 1u | ((wv_uni(reg.pre_status) >> 16) << 1); left; left &= left - 1, dst += 4)
                if (lane < 4) dst[lane] = c.gm[(u32)__builtin_ctz(left) * AVK_N_FIELDS + AVK_F_BP_TRUTH_TP + lane];
        }
        { /* SummaryWriter::add_comparison_benchmark (writers/summary.rs:146-163): the region's nonzero counters (a handful of
           * the 286) are added to the workgroup's tally in LDS, or — launches without a tail — straight to a partial tally */
            u64 *part_r = a.tally + (u64)((wave_id >> 2) % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE;
            for (u32 left = 1u | ((wv_uni(reg.pre_status) >> 16) << 1); left;) { /* only groups that can be nonzero (gm_step) */
                const u32 i = avk_opaque_u32(gm_step(left, lane));
                const u32 v = i < AVK_N_GROUPS * AVK_N_FIELDS ? c.gm[i] : 0u;
                if (v) {
                    if (wgt) avk_atomic_add_u32(wg_tally + i, v);
                    else avk_atomic_add_u64_global(part_r + i, v);
                }
            }
            /* the workgroup's LDS tally is 32 bits wide and a region can add 10^5 to a counter (BASEPAIR / RECORD_BP of 10 kbp alleles):
             * every wave moves the tally on to the 64-bit partial tally after 256 of its regions, long before a counter can wrap */
            if (wgt && (n_ok & 255u) == 0) {
                for (u32 i = lane; i < AVK_N_GROUPS * AVK_N_FIELDS; i += 64) {
                    const u32 v = avk_wg_xchg(wg_tally + i, 0u);
                    if (v) avk_atomic_add_u64_global(part_r + i, v);
                }
            }
        }
        if (a.seq_bytes && a.seq_len && reg.seq_stride) { /* SequenceBundle, waffle_solver.rs:237-246 */
            u8 *wn = c.pool + (u64)winner * c.nodeA_bytes;
            const HapPtr w0 = hap_ptr(wn + NODE_HDR, c.alw, c.wfcap, c.seqcap);
            const HapPtr w1 = hap_ptr(wn + NODE_HDR + c.hapA_bytes, c.alw, c.wfcap, c.seqcap);
            const HapHdr h0 = hap_load(w0.w), h1 = hap_load(w1.w);
            const u8 *src[5] = {c.ref, w0.tseq, w1.tseq, w0.qseq, w1.qseq};
            const u32 len[5] = {c.L, h0.t_len, h1.t_len, h0.q_len, h1.q_len};
            for (int k = 0; k < 5; ++k) {
                const u32 nbytes = len[k] < reg.seq_stride ? len[k] : reg.seq_stride;
                copy_bytes(a.seq_bytes + reg.seq_off + (u64)k * reg.seq_stride, src[k], nbytes);
                if (lane == 0) a.seq_len[5 * (u64)orig + k] = nbytes;
            }
        }
        wv_sync();
        AVK_RELEASE_SLOT()
#ifdef AVK_PHASE_TIMING
        if (lane == 0) {
            u64 *pc = a.tally + (u64)((wave_id >> 2) % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE + AVK_TALLY_LEN + 5;
            for (int k = 0; k < 6; ++k) avk_atomic_add_u64_global(pc + k, c.tphase[k]);
            avk_atomic_add_u64_global(pc + 6, avk_clock() - t_region0);
            if (a.group_metrics) /* profiling builds: searchA, searchB, metrics_setup, basepair ticks / 16 of this region in words -3 .. -6 (c.tphase is reset per region below) */
                for (int k = 1; k <= 4; ++k) a.group_metrics[(u64)orig * AVK_N_GROUPS * AVK_N_FIELDS + AVK_N_GROUPS * AVK_N_FIELDS - 2 - k] = (u32)(c.tphase[k] >> 4);
            if (a.group_metrics) { /* profiling builds: the region's own ticks and the tier that finished it in the last two words of its metric block (group 12 is never used) */
                a.group_metrics[(u64)orig * AVK_N_GROUPS * AVK_N_FIELDS + AVK_N_GROUPS * AVK_N_FIELDS - 1] = (u32)((avk_clock() - t_region0) >> 4);
                a.group_metrics[(u64)orig * AVK_N_GROUPS * AVK_N_FIELDS + AVK_N_GROUPS * AVK_N_FIELDS - 2] = (1u + a.pass_tier) | (a.high_priority ? 0x10u : 0u) | (a.work_list ? 0x20u : 0u) | (LAZY ? 0x40u : 0u) | (a.extra_n ? 0x80u : 0u);
            }
            avk_atomic_add_u64_global(pc + 7, 1);
            for (int k = 8; k < 16; ++k) avk_atomic_add_u64_global(pc + k, c.tphase[k]);
        }
#endif
    }

    if (wgt) {
        wv_sync();
        u32 gone = 0;
        if (lane == 0) {
            avk_wg_store(wg_ctl + 2 + wave_in_wg, 0xFFFFFFFFu); /* never parks again */
            gone = avk_wg_add(wg_ctl + 6, 1u) + 1u;
        }
        gone = wv_uni(wv_shfl(gone, 0));
        if (gone == n_wg_waves) { /* the last wave of the workgroup: its LDS adds and everybody else's are done */
            u64 *part_w = a.tally + (u64)((wave_id >> 2) % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE;
            for (u32 i = lane; i < AVK_N_GROUPS * AVK_N_FIELDS; i += 64) {
                const u32 v = wg_tally[i];
                if (v) avk_atomic_add_u64_global(part_w + i, v);
            }
        }
    }
    /* flush the private tally (SummaryWriter::add_comparison_benchmark, writers/summary.rs:146-163) into
     * one of the partial copies; avk_tally_reduce sums the copies */
    u64 *part = a.tally + (u64)((wave_id >> 2) % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE;
    if (lane == 0) {
        if (n_ok) avk_atomic_add_u64_global(part + AVK_TALLY_SOLVED, n_ok);
        if (n_err) avk_atomic_add_u64_global(part + AVK_TALLY_ERRORS, n_err);
        if (n_ok + n_err - n_cap - n_big) avk_atomic_add_u64_global(part + AVK_TALLY_LEN + tier, n_ok + n_err - n_cap - n_big);
        if (n_big) avk_atomic_add_u64_global(part + AVK_TALLY_LEN + (PASS_LDS ? 1 : 3), n_big);
        if (n_cap) avk_atomic_add_u64_global(part + AVK_TALLY_LEN + 4, n_cap);
    }
}

} // namespace avk
#endif
