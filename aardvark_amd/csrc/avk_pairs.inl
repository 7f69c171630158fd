/*
 * avk_pairs.inl — regions that hold ONE SNV on each side, the SAME one (position, ALT base): seven in ten regions of a whole-genome comparison.
 *
 * For such a region everything solve_compare_region (src/waffle_solver.rs:122-284) returns is decided by the two zygosities alone.  The four haplotype
 * strings are the window with or without one substituted base; a truth string and a query string are equal when both took the allele or neither did
 * and differ by exactly one substitution (edit distance 1, whatever the bases around it) otherwise, and which prefixes the search compares when
 * (order_variants, the sync points, query_optimizer.rs:258-265, :372-381) is a matter of positions, not of bases.  So costs, pop order, node ids, tied
 * optima, genotype flips and every metric are the same for all regions of one (truth zygosity, query zygosity) pair — 4 x 4 of them.
 *
 * Nothing of that is restated here.  The table is made by the SOLVER: pair_probe_records() builds one tile of sixteen such regions over a
 * synthetic window, the ordinary lane kernel (avk_lane.inl, class 0) solves it with the call's own configuration and its outputs redirected
 * into the table, and pair_worker() copies table rows: a region costs its 48-byte record, one word of the packed reference (the base under
 * the call must be an upper-case ACGT other than the ALT base — otherwise the region goes to the list of the wave-per-region kernels
 * like any region a lane hands back) and 24 bytes of output.  A zygosity pair whose probe did not come back Ok is handed over as well.
 * tests: every parity test that runs through the lane path runs through this one (most of their regions are of this kind);
 * tests/test_pairs.py compares the two paths on all sixteen pairs, lower-case / N / ALT == reference windows included.
 */
#ifndef AVK_PAIRS_INL
#define AVK_PAIRS_INL

#include "avk_dev_types.h"
#include "avk_wave.h"

namespace avk {
namespace pairs {

typedef uint32_t u32;
typedef uint64_t u64;

enum { N_SIG = 16, GM_WORDS = AVK_N_GROUPS * AVK_N_FIELDS };
enum { PROBE_REF_WORDS = 16, PROBE_L = 101, PROBE_POS = 50 };
#define AVK_PAIR_UNSET 0xFFFFFFFFu /* status word of a table row no probe has filled */

/* the table (device memory, one per context and max_branch_factor): outputs of the sixteen probe regions, in the layout the kernels write */
struct PairTable {
    u32 region[N_SIG][4];      /* status, ed_h1, ed_h2, n_optima | type_present << 16 */
    u32 var[N_SIG][2];         /* the truth call's word, the query call's word */
    u32 bp[N_SIG][8];          /* compact BASEPAIR groups: joint, SNV */
    u32 gm[N_SIG][GM_WORDS];   /* the full metric block */
};

struct PairArgs {
    const u32 *recs; /* the class's fast records (class AVK_FAST_PAIR: the record of a one-call class, AVK_FAST_WORDS_OF(1) words, tile-major) */
    u32 n_tiles;
    u32 gen_base;    /* record index (work order) of fast record 0 */
    const PairTable *tab;
    u32 *tile_counter;
};

AVK_TYPES_HD u32 pair_sig(u32 zt, u32 zq) { return (zt - AVK_ZYG_UNPHASED_HET) * 4u + (zq - AVK_ZYG_UNPHASED_HET); }

/* is this region of the class?  Both packers ask with the same words: one call per side, both SNVs with one base each way, same position, same ALT
 * base (2-bit code in the low bits of the packed allele), alt_ed 1 (REF base != ALT base), raw_space 1, zygosities that have alleles */
AVK_TYPES_HD bool pair_is_candidate(u32 tc, u32 qc, u32 t_pos, u32 q_pos, u32 t_a0, u32 t_a1, u32 q_a0, u32 q_a1, u32 t_type, u32 q_type, u32 t_zyg, u32 q_zyg,
                                    u32 t_alt_ed, u32 q_alt_ed, u32 t_raw, u32 q_raw, u32 t_alt2, u32 q_alt2) {
    return tc == 1 && qc == 1 && t_pos == q_pos && t_a0 == 1 && t_a1 == 1 && q_a0 == 1 && q_a1 == 1 && t_type == AVK_VT_SNV && q_type == AVK_VT_SNV &&
           t_zyg >= AVK_ZYG_UNPHASED_HET && t_zyg <= AVK_ZYG_HOM_ALT && q_zyg >= AVK_ZYG_UNPHASED_HET && q_zyg <= AVK_ZYG_HOM_ALT && t_alt_ed == 1 && q_alt_ed == 1 &&
           t_raw == 1 && q_raw == 1 && (t_alt2 & 3u) == (q_alt2 & 3u);
}

/* The probe: one tile of class-0 fast records (AVK_FAST_WORDS_OF(1) x 64 words), lanes 0..15 = the sixteen zygosity pairs over a window of
 * PROBE_L bases of the probe reference (all 'A': 2-bit zeros) with the SNV A>C at PROBE_POS, and the probe reference itself.  Plain host code. */
static inline void pair_probe_records(u32 *recs /* [AVK_FAST_WORDS_OF(1) * 64] */, u32 *ref2b /* [PROBE_REF_WORDS] */) {
    const u32 rw = AVK_FAST_WORDS_OF(1);
    for (u32 w = 0; w < rw * 64u; ++w) recs[w] = 0;
    for (u32 l = 0; l < 64; ++l) recs[1 * 64 + l] = 0xFFFFFFFFu; /* no region in this lane */
    for (u32 k = 0; k < PROBE_REF_WORDS; ++k) ref2b[k] = 0;
    for (u32 s = 0; s < N_SIG; ++s) {
        const u32 zt = AVK_ZYG_UNPHASED_HET + s / 4u, zq = AVK_ZYG_UNPHASED_HET + s % 4u;
        recs[0 * 64 + s] = 0;                                                      /* window starts at base 0 of the probe reference */
        recs[1 * 64 + s] = 0u | ((u32)PROBE_L << 4) | (1u << 12) | (1u << 14) | (2u << 16); /* shift 0, L, T 1, Q 1, order: depth 1 takes the query call */
        recs[2 * 64 + s] = 2u * s;                                                 /* per-call output words of probe s */
        recs[3 * 64 + s] = s;                                                      /* "caller index" = table row */
        for (u32 side = 0; side < 2; ++side) {
            u32 *v = recs + (AVK_FAST_HDR + 4u * side) * 64u + s;
            v[0] = (u32)PROBE_POS | (1u << 8) | (1u << 16) | ((u32)AVK_VT_SNV << 24) | ((side ? zq : zt) << 28);
            v[64] = 1u | (1u << 8); /* alt_ed 1, raw_space 1 */
            v[128] = 1u;            /* ALT allele "C" */
            v[192] = 0;
        }
    }
}

/* One persistent wave: claims PAIR_CLAIM tiles of 64 records at a time, every lane looks its regions up.  The tiles of a claim are handled together —
 * their record words are loaded first, then their reference words, then the table rows: a lookup is a chain of four memory round trips and nothing
 * else, so several of them in flight is all there is to gain.  `part` = the partial tally this wave adds to. */
enum { PAIR_CLAIM = 4 };
AVK_DEV void pair_worker(const AvkKernelArgs &a, const PairArgs &pa, u64 *part) {
    const u32 lane = (u32)wv_lane();
    const u32 rw = AVK_FAST_WORDS_OF(1);
    u32 cnt[N_SIG]; /* regions of each zygosity pair this wave finished (wave-uniform) */
#pragma unroll
    for (u32 s = 0; s < N_SIG; ++s) cnt[s] = 0;
    u32 n_ok = 0, n_err = 0;
    for (;;) {
        u32 t0 = 0;
        if (lane == 0) t0 = avk_atomic_add_u32_global(pa.tile_counter, (u32)PAIR_CLAIM);
        t0 = wv_uni(wv_shfl(t0, 0));
        if (t0 >= pa.n_tiles) break;
        u32 h0[PAIR_CLAIM], h1[PAIR_CLAIM], v_off[PAIR_CLAIM], orig[PAIR_CLAIM], tw[PAIR_CLAIM], qw[PAIR_CLAIM], alt[PAIR_CLAIM];
#pragma unroll
        for (u32 k = 0; k < PAIR_CLAIM; ++k) {
            const u32 t = t0 + k < pa.n_tiles ? t0 + k : t0; /* past the last tile: tile t0 again, not used */
            const u32 *rec = pa.recs + (u64)t * rw * 64u + lane;
            h0[k] = rec[0], h1[k] = rec[64], v_off[k] = rec[2 * 64], orig[k] = rec[3 * 64];
            tw[k] = rec[AVK_FAST_HDR * 64], qw[k] = rec[(AVK_FAST_HDR + 4) * 64], alt[k] = rec[(AVK_FAST_HDR + 2) * 64] & 3u;
            if (t0 + k >= pa.n_tiles) h1[k] = 0xFFFFFFFFu;
        }
        u32 refw[PAIR_CLAIM], excw[PAIR_CLAIM], sh[PAIR_CLAIM];
#pragma unroll
        for (u32 k = 0; k < PAIR_CLAIM; ++k) {
            const bool on = h1[k] != 0xFFFFFFFFu;
            const u64 base = on ? (u64)h0[k] * 16u + (h1[k] & 15u) + (tw[k] & 0xFFu) : 0ull; /* the base under the call, counted from the start of the packed reference */
            const u64 w = base >> 4;
            refw[k] = a.ref_2bit[w];
            excw[k] = (a.ref_exc[w >> 5] >> (w & 31)) & 1u;
            sh[k] = 2u * (u32)(base & 15u);
        }
#pragma unroll
        for (u32 k = 0; k < PAIR_CLAIM; ++k) {
            u32 sig = 0xFFu; /* 0xFF: nothing to count */
            if (h1[k] != 0xFFFFFFFFu) {
                const u32 refb = (refw[k] >> sh[k]) & 3u;
                const u32 s = pair_sig((tw[k] >> 28) & 7u, (qw[k] >> 28) & 7u);
                const avk_u4 r4 = *(const avk_u4 *)pa.tab->region[s];
                const u32 st = r4.x;
                if (excw[k] || refb == alt[k] || st == AVK_PAIR_UNSET) { /* not decided by the zygosities alone: a wave-per-region kernel solves it */
                    const u32 slot_o = avk_atomic_add_u32_global(a.overflow_count, 1u);
                    a.overflow_list[slot_o] = pa.gen_base + ((t0 + k) * 64u + lane);
                } else {
                    *(avk_u4 *)(a.region_out + 4 * (u64)orig[k]) = r4;
                    if (st == AVK_ST_OK) {
                        a.var_out[v_off[k]] = pa.tab->var[s][0];
                        a.var_out[v_off[k] + 1] = pa.tab->var[s][1];
                        sig = s;
                        n_ok += 1;
                    } else {
                        a.var_out[v_off[k]] = 0;
                        a.var_out[v_off[k] + 1] = 0;
                        n_err += 1;
                    }
                    if (a.group_metrics) {
                        u32 *g = a.group_metrics + (u64)orig[k] * GM_WORDS;
                        for (u32 i = 0; i < GM_WORDS; ++i) g[i] = st == AVK_ST_OK ? pa.tab->gm[s][i] : 0u;
                    }
                    if (a.bp_out) {
                        const u32 b0 = a.bp_off[orig[k]], b1 = a.bp_off[orig[k] + 1];
                        for (u32 i = 0; i < 4u * (b1 - b0) && i < 8u; ++i) a.bp_out[4 * (u64)b0 + i] = st == AVK_ST_OK ? pa.tab->bp[s][i] : 0u;
                    }
                }
            }
#pragma unroll
            for (u32 s = 0; s < N_SIG; ++s) cnt[s] += (u32)avk_popc64(wv_ballot(sig == s));
        }
    }
    /* the tally: counter i of the job += sum over the pairs of (regions of the pair) x (counter i of the pair's block) */
    for (u32 i = lane; i < GM_WORDS; i += 64) {
        u64 v = 0;
#pragma unroll
        for (u32 s = 0; s < N_SIG; ++s) v += (u64)cnt[s] * pa.tab->gm[s][i];
        if (v) avk_atomic_add_u64_global(part + i, v);
    }
    n_ok = wv_sum_u32(n_ok);
    n_err = wv_sum_u32(n_err);
    if (lane == 0) {
        if (n_ok) avk_atomic_add_u64_global(part + AVK_TALLY_SOLVED, n_ok);
        if (n_err) avk_atomic_add_u64_global(part + AVK_TALLY_ERRORS, n_err);
        if (n_ok + n_err) avk_atomic_add_u64_global(part + AVK_TALLY_LANE_SOLVED, n_ok + n_err);
    }
}

} // namespace pairs
} // namespace avk
#endif
