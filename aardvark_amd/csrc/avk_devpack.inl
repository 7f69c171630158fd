/*
 * avk_devpack.inl — the batch packer ON THE DEVICE.
 *
 * avk_pack.h (host threads) turns a caller batch (avk_region_batch, the fields of the reference's CompareRegion / Variant,
 * src/data_types/compare_region.rs:13-26, variants.rs:73-91) into what the solver kernels read: validation, Variant::alt_ed, the lane
 * classes and their cost keys, the work plan, the work order, fast records, region records and blobs.  On a whole genome that was
 * 12 + 18 ms of 16 host threads per call — nine times the solver kernels — and it cannot scale when eight ranks share one host.
 * Here the caller's arrays are copied to HBM AS THEY ARE (DMA from pinned memory, no host pass) and every step above is a kernel:
 *
 *   dp_variant        per call: alt_ed (prefix / suffix strip, then a 64-bit bit-vector edit distance), ALT packed 2 bits per base
 *   dp_region         per region: the validation rules of pack_batch, sizes, the lane class + cost key, the predicted workspace class
 *   dp_scan3          three exclusive prefix sums over the regions in caller order: per-call output words, blob words, sequence slots
 *   dp_lane_switch    which lane classes are large enough for a launch (plan_work_order's rule)
 *   dp_hist / dp_bucket_bases / dp_scatter   counting sort by (class, key): the work order
 *   dp_fast_records   tile-major fast records of the lane classes (avk_dev_types.h)
 *   dp_region_records AvkDevRegion + blob of every region, in work order (small regions: one lane each; large ones: one wave each)
 *   dp_unpack         after the solve: per-region records and per-call words into the caller's structure-of-arrays layout
 *
 * Every function below is written against avk_wave.h so that tests/emu runs the SAME code on the CPU and compares its output with
 * avk_pack.h's, record by record (tests/test_devpack.py).  The rules are pack_batch's, line for line; avk_pack.h stays the readable
 * statement of them and the host-side path (context option device_pack = 0).
 */
#ifndef AVK_DEVPACK_INL
#define AVK_DEVPACK_INL

#include "avk_dev_types.h"
#include "avk_wave.h"
#include "avk_pairs.inl"

namespace avk {
namespace dp {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

enum { DP_NB = (3 + AVK_FAST_CLASSES) * 256 }; /* buckets of the work order: class C, class B, bulk, lane classes 4..0, 256 keys each */
enum { DP_ERR_RANGE = 1, DP_ERR_BLOB = 2, DP_ERR_ALLELE = 4, /* batch-level errors (pack_batch returns AVK_E_ARG) */
       DP_NOTE_OUTSIDE = 256 };                              /* not an error: a region owns calls outside [v_lo, v_hi), the range the host guessed from the first and last region */
enum { DP_VF_BAD_RANGE = 1, DP_VF_ACGT32 = 2, DP_VF_PENDING = 4 };
enum { DP_NEED_BUCKETS = 9 };

/* the caller's arrays, as they are, in HBM (names of avk_region_batch) */
struct DpIn {
    const u32 *contig_idx; /* may be NULL */
    const u64 *start, *end, *t_off, *q_off;
    const u32 *t_cnt, *q_cnt;
    const u64 *var_pos;
    const u8 *var_type, *var_zyg;
    const u32 *var_raw; /* may be NULL */
    const u64 *a0_off, *a1_off;
    const u32 *a0_len, *a1_len;
    const u8 *alleles;
    u64 n_regions, n_variants, alleles_len;
    const u64 *contig_base, *contig_len;
    u32 n_contigs;
    u32 pairs_mode;
    u64 v_lo, v_hi; /* the calls the batch's regions are expected to own (results are copied back for this range only: batches may share call arrays) */
    u8 *owned;      /* [v_hi - v_lo] or NULL: forms with explicit call offsets — dp_region marks the calls its region owns, so that "every call of the range is owned
                       exactly once" is counted, not assumed from the sum of the counts (two regions may share a call while another call belongs to nobody) */
    /* The PACKED source (round 6): a batch that came as avk_packed_batch is read AS IT CAME — pk_start != NULL, the wide arrays above are then NULL and never made
     * (the widening pass wrote 44 B per region and 38 B per call that three later passes read back: 0.6 GB of a whole-genome call's 2 GB of packing traffic).
     * Every offset is implied by order: pk_voff / pk_aoff are the two exclusive prefix sums (calls before region r, allele bytes before call v); a call's position
     * is relative to its region's start, so whoever asks for it names the region's start. */
    const u32 *pk_start;
    const uint16_t *pk_len, *pk_contig, *pk_rel;
    const u8 *pk_tc, *pk_qc, *pk_tz, *pk_a0, *pk_a1;
    const u64 *pk_voff, *pk_aoff;
    AVK_DEV_MEMBER u32 contig_of(u64 r) const { return pk_start ? (pk_contig ? (u32)pk_contig[r] : 0u) : (contig_idx ? contig_idx[r] : 0u); }
    AVK_DEV_MEMBER u64 start_of(u64 r) const { return pk_start ? (u64)pk_start[r] : start[r]; }
    AVK_DEV_MEMBER u64 end_of(u64 r) const { return pk_start ? (u64)pk_start[r] + pk_len[r] : end[r]; }
    AVK_DEV_MEMBER u32 t_cnt_of(u64 r) const { return pk_start ? (u32)pk_tc[r] : t_cnt[r]; }
    AVK_DEV_MEMBER u32 q_cnt_of(u64 r) const { return pk_start ? (u32)pk_qc[r] : q_cnt[r]; }
    AVK_DEV_MEMBER u64 t_off_of(u64 r) const { return pk_start ? pk_voff[r] : t_off[r]; }
    AVK_DEV_MEMBER u64 q_off_of(u64 r) const { return pk_start ? pk_voff[r] + pk_tc[r] : q_off[r]; }
    AVK_DEV_MEMBER u64 pos_of(u64 v, u64 region_start) const { return pk_start ? region_start + pk_rel[v] : var_pos[v]; }
    AVK_DEV_MEMBER u32 a0_len_of(u64 v) const { return pk_start ? (u32)pk_a0[v] : a0_len[v]; }
    AVK_DEV_MEMBER u32 a1_len_of(u64 v) const { return pk_start ? (u32)pk_a1[v] : a1_len[v]; }
    AVK_DEV_MEMBER u64 a0_off_of(u64 v) const { return pk_start ? pk_aoff[v] : a0_off[v]; }
    AVK_DEV_MEMBER u64 a1_off_of(u64 v) const { return pk_start ? pk_aoff[v] + pk_a0[v] : a1_off[v]; }
    AVK_DEV_MEMBER u32 type_of(u64 v) const { return pk_start ? (u32)(pk_tz[v] & 15u) : (u32)var_type[v]; }
    AVK_DEV_MEMBER u32 zyg_of(u64 v) const { return pk_start ? (u32)(pk_tz[v] >> 4) : (u32)var_zyg[v]; }
    AVK_DEV_MEMBER u32 raw_of(u64 v, u32 l0, u32 l1) const { return var_raw ? var_raw[v] : (l0 > l1 ? l0 : l1); }
};

/* context options the plan depends on (plan_work_order's arguments) */
struct DpOpts {
    u64 tier0_bytes, tier1_bytes;
    u32 tier0_ed_cap, tier1_ed_cap;
    u32 solo_min_variants, max_branch, class_c_nodes_x2, lane_max_calls, lane_max_est;
    u32 head_est, het_min; /* het_min: regions with at least this many unphased heterozygous calls are big phasing searches (AVK_HET_SEARCH_MIN; 0 = no such rule) */
/* regions with at least this many estimated edits form the head of their lane class (1..15) */
    u32 lane_pairs; /* 1: regions with the same SNV on both sides get the class of their own (avk_pairs.inl) */
    u32 stripe_w; /* claim width the heads of the lane classes are dealt out over (avk_stripe_slot; 0 = sorted order) */
    u64 lane_min_regions; /* 0xFFFFFFFF = no lane classes */
    u64 lane_min_batch;
    u64 class_c_below; /* a batch with lane launches and at most this many regions outside them plans those regions as class C (0 = no such rule) */
};

/* per call, written by dp_variant */
struct DpVarInfo {
    u32 alt_ed, flags, a1lo, a1hi;
};

/* per region, written by dp_region (64 bytes: two sectors, read whole by the record writers) */
struct DpRegionInfo {
    u32 pre_status; /* as AvkDevRegion::pre_status */
    u32 len;
    u32 alle_bytes, blob_bytes, grow, ed_bound, seq_stride;
    u32 keys; /* fast_class | fast_key << 8 | plan class (0 C, 1 B, 2 bulk) << 16 | min(N, 255) << 24 */
    u64 ref_off;
    u32 bucket; /* final bucket of the work order (dp_hist) */
    u32 counts; /* candidates of the lane classes: truth calls | query calls << 8 (what dp_fast_record needs of the caller's region arrays, ... */
    u32 t_first, q_first; /* ... with the index of the side's first call) */
    u32 pad_[2];
};
/* per call of a candidate of the lane classes, written by dp_region: the call slot of its fast record as it will be stored (avk_dev_types.h) — dp_fast_record, which
 * visits the regions in work order, then reads 16 bytes per call instead of eight arrays of the caller's */
struct DpSlot {
    u32 w[4];
};

/* small shared state of one packing run */
struct DpState {
    u32 err;        /* DP_ERR_* */
    u32 n_pending;  /* calls whose alt_ed is left to the host (both stripped alleles longer than 64 symbols) */
    u32 lane_on[AVK_FAST_CLASSES];
    u32 lanes_any; /* some lane class has launches */
    u32 few_outside; /* ... and the regions outside them are few (class_c_below): those the wide kernel can take are class C */
    u64 have[AVK_FAST_CLASSES]; /* regions eligible per lane class */
    u64 total_v, total_blob_words, total_seq, total_groups; /* totals of the four scans */
    u64 need_hist[DP_NEED_BUCKETS];           /* class C regions by predicted HBM workspace: bucket b = at most 1 MB << b (the last one: more) */
    u32 hist[DP_NB];
    u32 base[DP_NB + 1];
    u32 cursor[DP_NB];
    /* the plan (WorkPlan of avk_pack.h) and the geometry of the fast records */
    u32 n_hbm, n_hard, n_fast_total;
    u32 n_hbm_notwide; /* class C regions (by size) that are not for avk_wide.inl by their record (avk_wide_static_ok): how the class's launches share the waves */
    u32 n_fast[AVK_FAST_CLASSES], n_fast_heavy[AVK_FAST_CLASSES], fast_base[AVK_FAST_CLASSES], fast_tiles[AVK_FAST_CLASSES], tile_first[AVK_FAST_CLASSES];
    u32 pad2_;
    u32 head_slots[AVK_FAST_CLASSES], pad4_; /* striped head of each lane class (avk_head_slots) */
    u64 fast_word_base[AVK_FAST_CLASSES], fast_words;
    u32 n_big, n_owned; /* regions left to the wave-per-region record writer; marked calls of DpIn::owned */
};

static_assert(DP_NB <= 65536, "DpArgs::bucket16 holds a bucket in two bytes");
struct DpArgs {
    DpIn in;
    DpOpts opt;
    DpVarInfo *vinfo;
    DpRegionInfo *rinfo;
    DpState *st;
    u32 *pending;       /* [n_variants] list of calls left to the host */
    u32 *v_off;         /* [n_regions] first per-call output word */
    u32 *blob_off8;     /* [n_regions] blob offset in units of 8 bytes */
    u64 *seq_off;       /* [n_regions] */
    u32 *bp_off;        /* [n_regions + 1] first compact BASEPAIR group of a region (1 + its call types groups each; none for regions that fail validation) */
    u32 *order;         /* [n_regions] work order: record k holds region order[k] */
    u32 *big_list;      /* [n_regions] work-order indices of regions with more than DP_SMALL_N calls */
    DpSlot *slots;      /* [n_variants] */
    uint16_t *bucket16; /* [n_regions] a region's bucket of the work order, from dp_hist to dp_scatter (two bytes a region instead of a word inside the 64-byte
                           region info: the two passes moved 230 MB each for it) */
    /* outputs */
    AvkDevRegion *regions;
    u32 *blob;
    u32 *fast;
};

#define DP_SMALL_N 48u /* regions with at most this many calls are written by one lane; larger ones by a whole wave */

/* ---- dp_widen: a batch in the compact form (avk_compact_batch, include/aardvark_amd.h) into the arrays of DpIn, in HBM ------------------------------ */
struct DpCompact {
    const u32 *contig_idx, *start, *len, *v_off;
    const uint16_t *t_cnt, *q_cnt;
    const u32 *var_pos, *a_off, *a0_len, *a1_len, *var_raw;
    const u8 *var_type_zyg;
    u64 n_regions, n_variants;
    /* the wide arrays (device), written here */
    u32 *w_contig, *w_t_cnt, *w_q_cnt, *w_a0_len, *w_a1_len, *w_raw;
    u64 *w_start, *w_end, *w_t_off, *w_q_off, *w_pos, *w_a0_off, *w_a1_off;
    u8 *w_type, *w_zyg;
};
AVK_DEV void dp_widen(const DpCompact &c, u64 i) { /* one lane: region i and call i */
    if (i < c.n_regions) {
        const u64 st = c.start[i], vo = c.v_off[i];
        const u32 tc = c.t_cnt[i], qc = c.q_cnt[i];
        if (c.w_contig) c.w_contig[i] = c.contig_idx[i];
        c.w_start[i] = st;
        c.w_end[i] = st + c.len[i];
        c.w_t_off[i] = vo;
        c.w_q_off[i] = vo + tc;
        c.w_t_cnt[i] = tc;
        c.w_q_cnt[i] = qc;
    }
    if (i < c.n_variants) {
        const u64 ao = c.a_off[i];
        const u32 l0 = c.a0_len[i], tz = c.var_type_zyg[i];
        c.w_pos[i] = c.var_pos[i];
        c.w_a0_off[i] = ao;
        c.w_a1_off[i] = ao + l0;
        c.w_a0_len[i] = l0;
        c.w_a1_len[i] = c.a1_len[i];
        if (c.w_raw) c.w_raw[i] = c.var_raw[i];
        c.w_type[i] = (u8)(tz & 15u);
        c.w_zyg[i] = (u8)(tz >> 4);
    }
}

/* ---- the packed form (avk_packed_batch): every offset implied by order; v_off / a_off come from two prefix sums on the device ------------------------- */
struct DpPacked {
    const uint16_t *contig_idx, *len, *rel_pos;
    const u32 *start, *var_raw;
    const u8 *t_cnt, *q_cnt, *var_type_zyg, *a0_len, *a1_len;
    const u64 *v_off, *a_off; /* exclusive prefix sums: calls before region r, allele bytes before call v */
    u64 n_regions, n_variants;
    u32 *w_contig, *w_t_cnt, *w_q_cnt, *w_a0_len, *w_a1_len, *w_raw;
    u64 *w_start, *w_end, *w_t_off, *w_q_off, *w_pos, *w_a0_off, *w_a1_off;
    u8 *w_type, *w_zyg;
};
AVK_DEV void dp_widen_packed(const DpPacked &c, u64 i) { /* one lane: region i (and the positions of its calls) and call i */
    if (i < c.n_regions) {
        const u64 st = c.start[i], vo = c.v_off[i];
        const u32 tc = c.t_cnt[i], qc = c.q_cnt[i];
        if (c.w_contig) c.w_contig[i] = c.contig_idx[i];
        c.w_start[i] = st;
        c.w_end[i] = st + c.len[i];
        c.w_t_off[i] = vo;
        c.w_q_off[i] = vo + tc;
        c.w_t_cnt[i] = tc;
        c.w_q_cnt[i] = qc;
        for (u32 k = 0; k < tc + qc && vo + k < c.n_variants; ++k) c.w_pos[vo + k] = st + c.rel_pos[vo + k];
    }
    if (i < c.n_variants) {
        const u64 ao = c.a_off[i];
        const u32 l0 = c.a0_len[i], l1 = c.a1_len[i], tz = c.var_type_zyg[i];
        c.w_a0_off[i] = ao;
        c.w_a1_off[i] = ao + l0;
        c.w_a0_len[i] = l0;
        c.w_a1_len[i] = l1;
        if (c.w_raw) c.w_raw[i] = c.var_raw ? c.var_raw[i] : (l0 > l1 ? l0 : l1);
        c.w_type[i] = (u8)(tz & 15u);
        c.w_zyg[i] = (u8)(tz >> 4);
    }
}

/* the packed MultiRegions (avk_packed_multi_batch): in_off = exclusive prefix sum of in_cnt over (region, input) */
struct DpPackedMulti {
    const uint16_t *contig_idx, *len, *rel_pos;
    const u32 *start, *var_raw;
    const u8 *in_cnt, *var_type_zyg, *a0_len, *a1_len;
    const u64 *in_off, *a_off;
    u64 n_multi, n_variants;
    u32 k;
    u32 *w_contig, *w_in_cnt, *w_a0_len, *w_a1_len, *w_raw;
    u64 *w_start, *w_end, *w_pos, *w_a0_off, *w_a1_off;
    u8 *w_type, *w_zyg;
};
AVK_DEV void dp_widen_packed_multi(const DpPackedMulti &c, u64 i) { /* one lane: MultiRegion i (and the positions of its calls) and call i */
    if (i < c.n_multi) {
        const u64 st = c.start[i], vo = c.in_off[i * c.k];
        if (c.w_contig) c.w_contig[i] = c.contig_idx[i];
        c.w_start[i] = st;
        c.w_end[i] = st + c.len[i];
        u32 calls = 0;
        for (u32 j = 0; j < c.k; ++j) {
            const u32 cnt = c.in_cnt[i * c.k + j];
            c.w_in_cnt[i * c.k + j] = cnt;
            calls += cnt;
        }
        for (u32 q = 0; q < calls && vo + q < c.n_variants; ++q) c.w_pos[vo + q] = st + c.rel_pos[vo + q];
    }
    if (i < c.n_variants) {
        const u64 ao = c.a_off[i];
        const u32 l0 = c.a0_len[i], l1 = c.a1_len[i], tz = c.var_type_zyg[i];
        c.w_a0_off[i] = ao;
        c.w_a1_off[i] = ao + l0;
        c.w_a0_len[i] = l0;
        c.w_a1_len[i] = l1;
        if (c.w_raw) c.w_raw[i] = c.var_raw ? c.var_raw[i] : (l0 > l1 ? l0 : l1);
        c.w_type[i] = (u8)(tz & 15u);
        c.w_zyg[i] = (u8)(tz >> 4);
    }
}

/* ---- the merge path on the device (solve_merge_region, src/merge_solver.rs:110-200) ------------------------------------------------------------------- */
/* a batch of MultiRegions (avk_multi_batch) as one CompareRegion-shaped item per input pair (i < j, lexicographic): input i plays the truth side */
struct DpPairs {
    const u32 *contig_idx; /* may be NULL */
    const u64 *start, *end, *in_off;
    const u32 *in_cnt;
    u64 n_multi;
    u32 k, ppr; /* inputs per region, pairs per region */
    u32 *w_contig, *w_t_cnt, *w_q_cnt;
    u64 *w_start, *w_end, *w_t_off, *w_q_off;
};
AVK_DEV void dp_expand_pairs(const DpPairs &c, u64 p) {
    if (p >= c.n_multi * c.ppr) return;
    const u64 m = p / c.ppr;
    u32 rest = (u32)(p % c.ppr), i = 0;
    while (rest >= c.k - 1 - i) rest -= c.k - 1 - i, ++i;
    const u32 j = i + 1 + rest;
    if (c.w_contig) c.w_contig[p] = c.contig_idx[m];
    c.w_start[p] = c.start[m];
    c.w_end[p] = c.end[m];
    c.w_t_off[p] = c.in_off[m * c.k + i];
    c.w_t_cnt[p] = c.in_cnt[m * c.k + i];
    c.w_q_off[p] = c.in_off[m * c.k + j];
    c.w_q_cnt[p] = c.in_cnt[m * c.k + j];
}
/* solve_merge_region's decision on top of the pair matrix (merge_solver.rs:149-199) for ONE region with at most KMAX inputs: pair(p, st, ex) hands out status
 * and exact-match flag of the region's pair p (lexicographic order).  The library's host function avk_merge_classify (KMAX 64) and the kernel behind
 * avk_merge_batch (KMAX 8) both call this. */
template <int KMAX, class PairFn>
AVK_HD void merge_classify_one(u32 k, const u32 *cnt, bool has_unknown, PairFn pair, u32 no_conflict_enabled, u32 majority_voting_enabled, int32_t conflict_selection,
                               int32_t *status, u8 *classification, u64 *members) {
    *status = 0;
    *classification = AVK_MERGE_DIFFERENT;
    *members = 0;
    if (has_unknown) { /* variant_delta_length bails on an Unknown zygosity before anything else */
        *status = AVK_ST_BAD_ZYGOSITY;
        return;
    }
    bool all_identical = true, no_conflict = true;
    u64 match[KMAX];
    for (u32 i = 0; i < k && i < (u32)KMAX; ++i) match[i] = 1ull << i;
    u64 p = 0;
    for (u32 i = 0; i < k; ++i)
        for (u32 j = i + 1; j < k; ++j, ++p) {
            int32_t pst;
            bool ex;
            pair(p, pst, ex);
            if (pst != 0) {
                *status = pst;
                return;
            }
            all_identical = all_identical && ex;
            no_conflict = no_conflict && (cnt[i] == 0 || cnt[j] == 0 || ex); /* :160-164 */
            if (ex) {
                match[i] |= 1ull << j;
                match[j] |= 1ull << i;
            }
        }
    const u32 majority = k / 2 + 1;
    u64 first_majority = 0;
    for (u32 i = 0; i < k && !first_majority; ++i) {
        u32 pc = 0;
        for (u64 x = match[i]; x; x &= x - 1) ++pc;
        if (pc >= majority) first_majority = match[i];
    }
    if (all_identical) *classification = AVK_MERGE_IDENTICAL;
    else if (no_conflict_enabled && no_conflict) {
        *classification = AVK_MERGE_NO_CONFLICT;
        for (u32 i = 0; i < k; ++i)
            if (cnt[i]) *members |= 1ull << i;
    } else if (majority_voting_enabled && first_majority) {
        *classification = AVK_MERGE_MAJORITY_AGREE;
        *members = first_majority;
    } else if (conflict_selection >= 0) {
        *classification = AVK_MERGE_CONFLICT_SELECTION;
        *members = (u64)conflict_selection;
    }
}
enum { DP_MERGE_KMAX = 8 }; /* regions with more inputs are classified by the host function */
struct DpMerge {
    const u32 *region_out; /* [n_multi * ppr][4] of the pair solve: status, exact */
    const u64 *in_off;
    const u32 *in_cnt;
    const u8 *var_zyg;
    u64 n_multi, n_variants;
    u32 k, ppr;
    u32 no_conflict_enabled, majority_voting_enabled;
    int32_t conflict_selection;
    int32_t *status;
    u8 *classification;
    u64 *members;
};
AVK_DEV void dp_merge_classify(const DpMerge &c, u64 m) {
    if (m >= c.n_multi) return;
    bool unknown = false;
    for (u32 i = 0; i < c.k; ++i) {
        const u64 off = c.in_off[m * c.k + i];
        const u32 cnt = c.in_cnt[m * c.k + i];
        for (u32 v = 0; v < cnt && off + v < c.n_variants; ++v) unknown = unknown || c.var_zyg[off + v] == AVK_ZYG_UNKNOWN;
    }
    const u32 *ro = c.region_out + 4 * (m * c.ppr);
    int32_t st;
    u8 cl;
    u64 mem;
    merge_classify_one<DP_MERGE_KMAX>(c.k, c.in_cnt + m * c.k, unknown, [ro](u64 p, int32_t &pst, bool &ex) {
        pst = (int32_t)ro[4 * p];
        ex = pst == 0 && ro[4 * p + 1] != 0;
    }, c.no_conflict_enabled, c.majority_voting_enabled, c.conflict_selection, &st, &cl, &mem);
    c.status[m] = st;
    c.classification[m] = cl;
    c.members[m] = mem;
}

/* ---- dp_variant: Variant::alt_ed (variants.rs:413-415 = wfa_ed(allele0, allele1), sequence_alignment.rs:9-13) ------------------------ */
/* Unit-cost edit distance of a pattern of at most 64 symbols against a text of any length: the bit-vector recurrence of Myers
 * (J. ACM 46, 1999) in Hyyrö's global-distance form (the horizontal delta of row 0 is +1 in every column).  Symbols are bytes, so
 * the match mask of a text symbol is built by comparing it with the pattern.  O(|text| * |pattern|) byte compares, no arrays. */
AVK_DEV u32 dp_myers64(const u8 *pat, u32 m, const u8 *txt, u64 n) {
    const u64 top = 1ull << (m - 1);
    u64 Pv = m == 64 ? ~0ull : ((1ull << m) - 1ull), Mv = 0;
    u32 score = m;
    for (u64 j = 0; j < n; ++j) {
        const u8 c = txt[j];
        u64 Eq = 0;
        for (u32 i = 0; i < m; ++i) Eq |= (u64)(pat[i] == c) << i;
        const u64 Xv = Eq | Mv;
        const u64 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        u64 Ph = Mv | ~(Xh | Pv);
        u64 Mh = Pv & Xh;
        score += (Ph & top) ? 1u : 0u;
        score -= (Mh & top) ? 1u : 0u;
        Ph = (Ph << 1) | 1ull;
        Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
    }
    return score;
}

AVK_DEV void dp_variant(const DpArgs &a, u64 v) {
    if (v >= a.in.n_variants) return;
    DpVarInfo o;
    o.alt_ed = 0, o.flags = 0, o.a1lo = 0, o.a1hi = 0;
    const u64 o0 = a.in.a0_off_of(v), o1 = a.in.a1_off_of(v);
    const u32 l0 = a.in.a0_len_of(v), l1 = a.in.a1_len_of(v);
    if (o0 + l0 > a.in.alleles_len || o1 + l1 > a.in.alleles_len || o0 + l0 < o0 || o1 + l1 < o1) {
        o.flags = DP_VF_BAD_RANGE;
        a.vinfo[v] = o;
        return;
    }
    const u8 *s0 = a.in.alleles + o0, *s1 = a.in.alleles + o1;
    if (l1 >= 1 && l1 <= 32) { /* ALT as the lanes read it: 2 bits per base, 16 bases per word (pack_bases_2bit) */
        bool ok = true;
        u64 w = 0;
        for (u32 i = 0; i < l1; ++i) {
            const u8 ch = s1[i];
            const u32 code = ch == 'C' ? 1u : (ch == 'G' ? 2u : (ch == 'T' ? 3u : 0u));
            ok = ok && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
            w |= (u64)code << (2 * i);
        }
        o.a1lo = (u32)w;
        o.a1hi = (u32)(w >> 32);
        if (ok) o.flags |= DP_VF_ACGT32;
    }
    /* host_edit_distance (avk_pack.h): common prefix and suffix first */
    u64 n = l0, m = l1;
    while (n && m && s0[0] == s1[0]) ++s0, ++s1, --n, --m;
    while (n && m && s0[n - 1] == s1[m - 1]) --n, --m;
    if (n == 0) o.alt_ed = (u32)m;
    else if (m == 0) o.alt_ed = (u32)n;
    else if (n == 1 && m == 1) o.alt_ed = 1;
    else if (m <= 64 && m <= n) o.alt_ed = dp_myers64(s1, (u32)m, s0, n);
    else if (n <= 64) o.alt_ed = dp_myers64(s0, (u32)n, s1, m);
    else { /* two long unrelated alleles: rare enough for the host (upload fills these in and the region pass runs again) */
        o.flags |= DP_VF_PENDING;
        const u32 k = avk_atomic_add_u32_global(&a.st->n_pending, 1u);
        a.pending[k] = (u32)v;
    }
    a.vinfo[v] = o;
}

/* ---- dp_region: pack_batch's validation and sizes, the lane class and its cost key, the predicted workspace class ------------------- */
struct DpCall { /* FastCall of avk_pack.h */
    u32 pos, a0, a1, alt_ed, type, zyg, lo, hi, raw;
};
AVK_DEV u32 dp_copies(u32 z) { return z == AVK_ZYG_HOM_ALT ? 2u : 1u; }
/* fast_cost_key (avk_pack.h), the same arithmetic on the packed ALT words.  c[0..2] = the truth calls, c[3..5] = the query calls; every index is a
 * compile-time constant (the loops are unrolled over the three slots of a side), so the six records stay in registers */
AVK_DEV u32 dp_cost_key(const DpCall (&c)[2 * AVK_FAST_MAXV], u32 tc, u32 qc) {
    u32 est = 0, used = 0, nhet = 0;
#pragma unroll
    for (u32 i = 0; i < AVK_FAST_MAXV; ++i) {
        if (i >= tc) continue;
        int m = -1;
#pragma unroll
        for (u32 j = 0; j < AVK_FAST_MAXV; ++j) {
            const DpCall &q = c[AVK_FAST_MAXV + j];
            if (j < qc && m < 0 && !((used >> j) & 1u) && c[i].pos == q.pos && c[i].a0 == q.a0 && c[i].a1 == q.a1 && c[i].lo == q.lo && c[i].hi == q.hi) m = (int)j;
        }
        if (m < 0) est += c[i].alt_ed * dp_copies(c[i].zyg);
        else {
            used |= 1u << m;
            const u32 qz = m == 0 ? c[AVK_FAST_MAXV].zyg : (m == 1 ? c[AVK_FAST_MAXV + 1].zyg : c[AVK_FAST_MAXV + 2].zyg);
            const u32 x = dp_copies(c[i].zyg), y = dp_copies(qz);
            est += c[i].alt_ed * (x > y ? x - y : y - x);
        }
        nhet += c[i].zyg != AVK_ZYG_HOM_ALT;
    }
#pragma unroll
    for (u32 j = 0; j < AVK_FAST_MAXV; ++j) {
        if (j >= qc) continue;
        const DpCall &q = c[AVK_FAST_MAXV + j];
        if (!((used >> j) & 1u)) est += q.alt_ed * dp_copies(q.zyg);
        nhet += q.zyg != AVK_ZYG_HOM_ALT;
    }
#pragma unroll
    for (u32 side = 0; side < 2; ++side) {
        const u32 n = side ? qc : tc;
        bool mixed = false;
        u32 longest = 0;
#pragma unroll
        for (u32 i = 0; i < AVK_FAST_MAXV; ++i) {
            if (i >= n) continue;
            mixed = mixed || c[side * AVK_FAST_MAXV + i].type != c[side * AVK_FAST_MAXV].type;
            longest = c[side * AVK_FAST_MAXV + i].alt_ed > longest ? c[side * AVK_FAST_MAXV + i].alt_ed : longest;
        }
        if (mixed && longest > 2) est += longest - 2;
    }
    const u32 n_all = tc + qc, lo = tc > qc ? tc : qc;
    const u32 extra = n_all - lo;
    return ((est > 15 ? 15u : est) << 4) | ((extra > 3 ? 3u : extra) << 2) | (nhet > 3 ? 3u : nhet);
}
/* plan_work_order's `need`: bytes a region's search is predicted to take in a workspace tier */
AVK_DEV u64 dp_need(u32 len, u32 t_cnt, u32 q_cnt, u32 ed_bound, u64 N, u64 alle, u64 grow, u32 tier_cap, u64 nodes, u32 max_branch) {
    const u64 seqcap = ((u64)len + grow + 7) & ~7ull;
    const u64 maxT = t_cnt > q_cnt ? t_cnt : q_cnt;
    const u64 alw = maxT ? (maxT + 63) >> 6 : 1;
    u64 cap = tier_cap;
    if (cap && ed_bound < cap) cap = ed_bound ? ed_bound : 1;
    u64 wfcap = cap ? 2 * cap + 2 : 2 * seqcap + 4;
    if (wfcap > 2 * seqcap + 4) wfcap = 2 * seqcap + 4;
    const u64 hapA = (48 + 16 * alw + 4 * wfcap + 2 * seqcap + 15) & ~15ull, nodeA = 16 + 2 * hapA;
    const u64 optcap = max_branch < 4096 ? max_branch : 4096;
    const u64 fixed = len + 8 + 28 * N + alle + 32 + 32 * N + 4 * N + 16 + 32 + 4 * optcap + 8 * 8 * alw + 8 * 4 * alw * 8 + 32 + 64;
    return fixed + nodes * (nodeA + 16);
}

/* returns the three quantities the scans add up (per-call words, blob words, sequence bytes) and the lane class the region is eligible for
 * (0 = none; the caller counts them into DpState::have — one atomic per wave, not per region) through the references */
AVK_DEV void dp_region(const DpArgs &a, u64 r, u32 &n_calls, u32 &blob_words, u64 &seq_bytes, u32 &lane_class, u32 &need_bucket, u32 &n_groups) {
    n_calls = 0, blob_words = 0, seq_bytes = 0, lane_class = 0, need_bucket = 0xFFu, n_groups = 0;
    if (r >= a.in.n_regions) return;
    const DpIn &in = a.in;
    DpRegionInfo ri;
    ri.pre_status = 0, ri.len = 0, ri.alle_bytes = 0, ri.blob_bytes = 0, ri.grow = 0, ri.ed_bound = 0, ri.seq_stride = 1, ri.keys = 2u << 16, ri.ref_off = 0, ri.bucket = 0, ri.counts = 0, ri.t_first = 0, ri.q_first = 0, ri.pad_[0] = ri.pad_[1] = 0;
    const u32 tc = in.t_cnt_of(r), qc = in.q_cnt_of(r);
    const u64 toff = in.t_off_of(r), qoff = in.q_off_of(r), nv = in.n_variants;
    if (toff > nv || (u64)tc > nv - toff || qoff > nv || (u64)qc > nv - qoff) { /* pass 1 of pack_batch: the batch is rejected */
        avk_atomic_or_u32_global(&a.st->err, DP_ERR_RANGE);
        a.rinfo[r] = ri;
        return;
    }
    const u64 N = (u64)tc + qc;
    if ((tc && (toff < in.v_lo || toff + tc > in.v_hi)) || (qc && (qoff < in.v_lo || qoff + qc > in.v_hi))) avk_atomic_or_u32_global(&a.st->err, DP_NOTE_OUTSIDE);
    if (in.owned) {
        for (u32 i = 0; i < tc; ++i)
            if (toff + i >= in.v_lo && toff + i < in.v_hi) in.owned[toff + i - in.v_lo] = 1;
        for (u32 i = 0; i < qc; ++i)
            if (qoff + i >= in.v_lo && qoff + i < in.v_hi) in.owned[qoff + i - in.v_lo] = 1;
    }
    const u32 c = in.contig_of(r);
    const u64 start = in.start_of(r), end = in.end_of(r);
    u32 pre = 0;
    if (c >= in.n_contigs || start > end || end > in.contig_len[c] || end - start > 0x7FFFFFFFull) pre = AVK_ST_INVALID_INPUT;
    if (N > 60000) pre = AVK_ST_INVALID_INPUT;
    ri.len = pre ? 0u : (u32)(end - start);
    ri.ref_off = pre ? 0ull : in.contig_base[c] + start;
    u64 alle = 0, g0 = 0, g1 = 0, ed_sum = 0;
    i64 delta_t = 0, delta_q = 0;
    u32 types = 0, zflags = 0, nhet_u = 0;
    bool bad_zyg = false, bad_allele = false;
    for (int side = 0; side < 2; ++side) {
        const u64 off = side == 0 ? toff : qoff;
        const u32 cnt = side == 0 ? tc : qc;
        u64 last = 0;
        for (u32 i = 0; i < cnt; ++i) {
            const u64 v = off + i, pos = in.pos_of(v, start);
            const u32 l0 = in.a0_len_of(v), l1 = in.a1_len_of(v);
            const u32 big = l0 > l1 ? l0 : l1;
            const u32 raw = in.raw_of(v, l0, l1);
            const u32 vt = in.type_of(v), zy = in.zyg_of(v);
            const DpVarInfo vi = a.vinfo[v];
            alle += (u64)l0 + l1;
            bad_allele = bad_allele || (vi.flags & DP_VF_BAD_RANGE);
            if (l0 == 0 || l1 == 0 || raw < big) pre = AVK_ST_INVALID_INPUT;
            if (vt >= AVK_N_VARIANT_TYPES || zy > AVK_ZYG_HOM_ALT) pre = AVK_ST_INVALID_INPUT;
            if (pos < start || pos + l0 > end || pos < last) pre = AVK_ST_INVALID_INPUT;
            last = pos;
            if (zy == AVK_ZYG_UNKNOWN || zy == AVK_ZYG_HOM_REF) bad_zyg = true;
            if (zy == AVK_ZYG_UNKNOWN) zflags |= 1u;
            if (zy == AVK_ZYG_HOM_REF) zflags |= 2u;
            nhet_u += zy == AVK_ZYG_UNPHASED_HET ? 1u : 0u;
            {
                const i64 w = zy == AVK_ZYG_HOM_ALT ? 2 : ((zy >= AVK_ZYG_UNPHASED_HET && zy <= AVK_ZYG_PHASED_HET10) ? 1 : 0);
                (side == 0 ? delta_t : delta_q) += ((i64)l1 - (i64)l0) * w;
            }
            if (l1 > l0) (side == 0 ? g0 : g1) += l1 - l0;
            if (vt < AVK_N_VARIANT_TYPES) types |= 1u << vt;
            ed_sum += vi.alt_ed;
        }
    }
    if (bad_allele) avk_atomic_or_u32_global(&a.st->err, DP_ERR_ALLELE);
    u64 bytes = 0;
    if (N <= 60000) bytes = ((N * sizeof(AvkBlobVar) + 15) & ~15ull) + ((alle + 15) & ~15ull) + N * sizeof(AvkOrdVar) + 32;
    if (bytes > 0x7FFFFFFFull) {
        avk_atomic_or_u32_global(&a.st->err, DP_ERR_BLOB);
        bytes = 0;
    }
    ri.alle_bytes = (u32)(alle < 0xFFFFFFFFull ? alle : 0xFFFFFFFFull);
    const u64 grow = g0 > g1 ? g0 : g1;
    if (!pre && bad_zyg) pre = AVK_ST_BAD_ZYGOSITY;
    if (!pre && grow > 0x7FFFFFFFull) pre = AVK_ST_INVALID_INPUT;
    { /* seq_stride_of */
        u64 s = (end >= start ? end - start : 0) + grow;
        if (s < 1) s = 1;
        if (s > 0xFFFFFFFFull) s = 0xFFFFFFFFull;
        ri.seq_stride = (u32)s;
    }
    u32 fast_class = 0, fast_key = 0;
    ri.pre_status = pre;
    if (!pre) {
        ri.blob_bytes = (u32)bytes;
        ri.grow = (u32)grow;
        ri.pre_status |= types << 16;
        ri.ed_bound = (u32)(ed_sum < 0x7FFFFFFFull ? ed_sum : 0x7FFFFFFFull);
        if (tc <= AVK_FAST_MAXV && qc <= AVK_FAST_MAXV && N >= 1 && ri.len <= 255 && ed_sum <= 255) {
            /* a candidate for the lanes: its (at most six) calls once more, into registers (the loop is unrolled over the slots: static indices), with
             * the classes' per-call limits (pack_batch checks them on the blob) */
            DpCall calls[2 * AVK_FAST_MAXV];
            bool lane_ok = true;
#pragma unroll
            for (u32 sidx = 0; sidx < 2 * AVK_FAST_MAXV; ++sidx) {
                const u32 side = sidx / AVK_FAST_MAXV, j = sidx % AVK_FAST_MAXV;
                DpCall &k = calls[sidx];
                k.pos = k.a0 = k.a1 = k.alt_ed = k.type = k.zyg = k.lo = k.hi = k.raw = 0;
                if (j >= (side ? qc : tc)) continue;
                const u64 v = (side ? qoff : toff) + j;
                const u32 l0 = in.a0_len_of(v), l1 = in.a1_len_of(v);
                const u32 raw = in.raw_of(v, l0, l1);
                const DpVarInfo vi = a.vinfo[v];
                const u64 rel = in.pos_of(v, start) - start;
                lane_ok = lane_ok && rel <= 255 && l0 <= 255 && l1 <= 32 && vi.alt_ed <= 255 && raw <= 0xFFFF && (vi.flags & DP_VF_ACGT32);
                k.pos = (u32)rel, k.a0 = l0, k.a1 = l1, k.alt_ed = vi.alt_ed, k.type = in.type_of(v), k.zyg = in.zyg_of(v), k.lo = vi.a1lo, k.hi = vi.a1hi, k.raw = raw;
            }
            for (int cl = 0; cl < AVK_FAST_GENERIC && lane_ok; ++cl) {
                const u32 W = AVK_FAST_CLASS[cl].W, maxv = AVK_FAST_CLASS[cl].maxv;
                if (tc <= maxv && qc <= maxv && (u64)ri.len + ri.grow <= 16ull * W) {
                    fast_class = (u32)cl + 1u;
                    fast_key = dp_cost_key(calls, tc, qc);
                    if ((fast_key >> 4) > a.opt.lane_max_est) fast_class = 0;
                    if (maxv > 2) fast_key = 0u; /* the three-call class keeps the caller's order (avk_pack.h) */
                    if (maxv > 2 && a.opt.het_min && nhet_u >= a.opt.het_min) fast_class = 0; /* a big phasing search (avk_dev_types.h): not for a lane */
                    break;
                }
            }
            /* the same SNV on both sides: looked up, not searched (avk_pairs.inl) */
            if (fast_class && a.opt.lane_pairs && !in.pairs_mode && /* (merge batches are solved in mode 1: nothing to look up) */
                pairs::pair_is_candidate(tc, qc, calls[0].pos, calls[AVK_FAST_MAXV].pos, calls[0].a0, calls[0].a1, calls[AVK_FAST_MAXV].a0, calls[AVK_FAST_MAXV].a1, calls[0].type,
                                         calls[AVK_FAST_MAXV].type, calls[0].zyg, calls[AVK_FAST_MAXV].zyg, calls[0].alt_ed, calls[AVK_FAST_MAXV].alt_ed, calls[0].raw,
                                         calls[AVK_FAST_MAXV].raw, calls[0].lo, calls[AVK_FAST_MAXV].lo)) {
                fast_class = (u32)AVK_FAST_PAIR + 1u;
                fast_key = 0;
            }
            if (fast_class) { /* the call slots of its fast record */
                ri.counts = tc | (qc << 8), ri.t_first = (u32)toff, ri.q_first = (u32)qoff;
#pragma unroll
                for (u32 sidx = 0; sidx < 2 * AVK_FAST_MAXV; ++sidx) {
                    const u32 side = sidx / AVK_FAST_MAXV, j = sidx % AVK_FAST_MAXV;
                    if (j >= (side ? qc : tc)) continue;
                    const DpCall &k = calls[sidx];
                    DpSlot sl;
                    sl.w[0] = k.pos | (k.a0 << 8) | (k.a1 << 16) | ((k.type & 0xFu) << 24) | ((k.zyg & 7u) << 28);
                    sl.w[1] = k.alt_ed | (k.raw << 8);
                    sl.w[2] = k.lo, sl.w[3] = k.hi;
                    *(avk_u4 *)&a.slots[(side ? qoff : toff) + j] = *(const avk_u4 *)&sl;
                }
            }
        }
    }
    if (in.pairs_mode) { /* solve_merge_region's pre-checks (merge_solver.rs:119-147, :211-223), as upload_internal applies them */
        if ((ri.pre_status & 0xFFFFu) != AVK_ST_INVALID_INPUT) {
            if (zflags & 1u) ri.pre_status = AVK_ST_BAD_ZYGOSITY;
            else if (delta_t != delta_q) ri.pre_status = AVK_PRE_SKIP_OK;
            else if (zflags & 2u) ri.pre_status = AVK_ST_BAD_ZYGOSITY;
            else ri.pre_status = 0;
        }
    }
    /* plan_work_order: the class of a region the lanes do not take */
    u32 cls = 2;
    const bool failed = (ri.pre_status & 0xFFFFu) != 0;
    /* a big phasing search (AVK_HET_SEARCH_MIN): class C — in batches that have lane launches, which is decided after this kernel (dp_bucket_of reads the flag) */
    const bool het_search = !failed && N != 0 && a.opt.solo_min_variants != 0 && a.opt.tier1_bytes && a.opt.het_min && nhet_u >= a.opt.het_min;
    if (!failed && N != 0 && a.opt.solo_min_variants != 0) {
        const bool c_by_size = a.opt.tier1_bytes && dp_need(ri.len, tc, qc, ri.ed_bound, N, ri.alle_bytes, ri.grow, a.opt.tier1_ed_cap, ((u64)a.opt.class_c_nodes_x2 * N + 1) / 2, a.opt.max_branch) > a.opt.tier1_bytes;
        if (c_by_size || het_search) {
            if (c_by_size) cls = 0;
            if (c_by_size && !avk_wide_static_ok(ri.len, ri.grow, ri.ed_bound, tc, qc, 0u)) avk_atomic_add_u32_global(&a.st->n_hbm_notwide, 1u); /* (rare, or the whole batch: a batch of large windows) */
            /* how large an HBM slice the region is predicted to want (no edit-distance cap there): the host sizes the per-wave slices of the batch's
             * launches by the distribution — large windows (--min-variant-gap 1000) outgrow the default 1 MB by the thousand, and the shared big
             * slices serialise whatever overflows */
            /* (priced with two wavefront entries per base although the tier sizes the fronts by the region's bound since round 3 — hbm_ed_cap: the search of a
             * large window keeps many more nodes alive than the 6 N counted here, and the generous slice is what holds them; priced by the real fronts, 417 instead
             * of 10 of 49 k large-window regions went on to the shared big slices and the step took 1.6 instead of 1.2 s) */
            const u64 need2 = dp_need(ri.len, tc, qc, ri.ed_bound, N, ri.alle_bytes, ri.grow, 0u, ((u64)a.opt.class_c_nodes_x2 * N + 1) / 2, a.opt.max_branch);
            u32 b = 0;
            while (b + 1 < DP_NEED_BUCKETS && need2 > (1ull << (20 + b))) ++b;
            need_bucket = b;
        }
        if (cls != 0 && (N >= a.opt.solo_min_variants || dp_need(ri.len, tc, qc, ri.ed_bound, N, ri.alle_bytes, ri.grow, a.opt.tier0_ed_cap, 2 * N + 1, a.opt.max_branch) > a.opt.tier0_bytes))
            cls = 1;
    }
    if (het_search) cls |= 0x80u;
    /* (class_c_below: decided after this kernel as well) */
    if (!failed && N != 0 && a.opt.solo_min_variants != 0 && a.opt.tier1_bytes && avk_wide_static_ok(ri.len, ri.grow, ri.ed_bound, tc, qc, 0u)) cls |= 0x40u;
    if (failed) fast_class = 0; /* `have` and the work order only count regions that will be solved */
    ri.keys = fast_class | (fast_key << 8) | (cls << 16) | ((u32)(N > 255 ? 255 : N) << 24);
    a.rinfo[r] = ri;
    lane_class = fast_class;
    if (!failed) { /* compact BASEPAIR groups: the joint one and one per call type of the region */
        n_groups = 1;
        for (u32 x = ri.pre_status >> 16; x; x &= x - 1) n_groups += 1;
    }
    n_calls = (u32)N;
    blob_words = ri.blob_bytes / 4;
    seq_bytes = 5ull * ri.seq_stride;
}

/* ---- dp_lane_switch: which lane classes get launches (plan_work_order), one thread ------------------------------------------------ */
AVK_DEV void dp_lane_switch(const DpArgs &a) {
    const u64 scale[AVK_FAST_CLASSES] = {1, 1, 16, 16, 2, 1};
    u64 have_all = 0;
    u32 on[AVK_FAST_CLASSES];
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) {
        const u64 h = a.st->have[fc];
        on[fc] = a.opt.lane_min_regions != 0xFFFFFFFFull && h > 0 && h >= a.opt.lane_min_regions * scale[fc] && AVK_FAST_CLASS[fc].maxv <= a.opt.lane_max_calls;
        have_all += on[fc] ? h : 0;
    }
    if (a.opt.lane_min_regions != 0 && have_all < a.opt.lane_min_batch)
        for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) on[fc] = 0;
    u32 any = 0;
    u64 have_on = 0;
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) a.st->lane_on[fc] = on[fc], any |= on[fc], have_on += on[fc] ? a.st->have[fc] : 0;
    a.st->lanes_any = any;
    a.st->few_outside = any && a.opt.class_c_below != 0 && a.in.n_regions - have_on <= a.opt.class_c_below;
}

/* the bucket of a region in the work order: [class C | class B | bulk | lane class 4 | .. | lane class 0], most expensive key first */
AVK_DEV u32 dp_bucket_of(const DpArgs &a, u64 r) {
    const u32 k = a.rinfo[r].keys;
    const u32 fc = k & 0xFFu;
    if (fc && a.st->lane_on[fc - 1]) return 256u * (3u + (AVK_FAST_CLASSES - fc)) + (255u - ((k >> 8) & 0xFFu));
    u32 cls = (k >> 16) & 0x3Fu;
    if (((k >> 16) & 0x80u) && a.st->lanes_any) cls = 0; /* a big phasing search, in a batch with lane launches: class C */
    if (((k >> 16) & 0x40u) && a.st->few_outside) cls = 0; /* few regions outside the lanes: class C, the wide kernel */
    return 256u * cls + (255u - (k >> 24));
}

/* where the region that came `pos`-th in the sort goes in the work order: the heads of the lane classes are striped (avk_stripe_slot) */
AVK_DEV u32 dp_order_slot(const DpArgs &a, u32 bucket, u32 pos) {
    if (bucket < 256u * 3u) return pos;
    const u32 fc = (u32)AVK_FAST_CLASSES - 1u - (bucket / 256u - 3u);
    const DpState &s = *a.st;
    return s.fast_base[fc] + avk_stripe_slot(pos - s.fast_base[fc], s.head_slots[fc], a.opt.stripe_w);
}

/* ---- dp_bucket_bases: exclusive scan of the histogram, the plan, the geometry of the fast records; one thread ---------------------- */
AVK_DEV void dp_bucket_plan(const DpArgs &a);
AVK_DEV void dp_bucket_bases(const DpArgs &a) {
    DpState &s = *a.st;
    u32 run = 0;
    for (u32 b = 0; b < DP_NB; ++b) {
        s.base[b] = run;
        s.cursor[b] = run;
        run += s.hist[b];
    }
    s.base[DP_NB] = run;
    dp_bucket_plan(a);
}
/* what follows from the bucket bases (the GPU makes the bases with a workgroup, avk_dp_bucket_bases_kernel, and calls this from one thread) */
AVK_DEV void dp_bucket_plan(const DpArgs &a) {
    DpState &s = *a.st;
    s.n_hbm = s.base[256] - s.base[0];
    s.n_hard = s.base[512] - s.base[256];
    s.n_fast_total = 0;
    u32 tiles = 0;
    u64 words = 0;
    for (int fc = AVK_FAST_CLASSES - 1; fc >= 0; --fc) {
        const u32 seg = 3u + (AVK_FAST_CLASSES - 1 - fc);
        s.fast_base[fc] = s.base[256 * seg];
        s.n_fast[fc] = s.base[256 * (seg + 1)] - s.base[256 * seg];
        s.n_fast_heavy[fc] = s.base[256 * seg + 256 - 16 * (a.opt.head_est ? a.opt.head_est : 1u)] - s.base[256 * seg]; /* estimated edits (cost key >> 4) >= head_est <=> sort key below 256 - 16 head_est */
        s.head_slots[fc] = avk_head_slots(AVK_FAST_CLASS[fc].maxv, s.n_fast[fc], s.n_fast_heavy[fc], a.opt.stripe_w);
        s.n_fast_total += s.n_fast[fc];
        s.tile_first[fc] = tiles;
        s.fast_word_base[fc] = words;
        s.fast_tiles[fc] = (s.n_fast[fc] + 63u) / 64u;
        tiles += s.fast_tiles[fc];
        words += (u64)s.fast_tiles[fc] * AVK_FAST_WORDS_OF(AVK_FAST_CLASS[fc].maxv) * 64u;
    }
    s.fast_words = words;
}

/* ---- dp_fast_record: one record slot (class fc, tile, lane), build_fast_records of avk_pack.h ------------------------------------- */
AVK_DEV void dp_fast_record(const DpArgs &a, u32 fc, u32 tile_in_class, u32 lane) {
    const DpState &s = *a.st;
    const u32 maxv = AVK_FAST_CLASS[fc].maxv, rw = AVK_FAST_WORDS_OF(maxv);
    u32 *T = a.fast + s.fast_word_base[fc] + (u64)tile_in_class * rw * 64u + lane;
    const u32 k = tile_in_class * 64u + lane;
    if (k >= s.n_fast[fc]) {
        for (u32 w = 0; w < rw; ++w) T[w * 64] = w == 1 ? 0xFFFFFFFFu : 0u;
        return;
    }
    const u32 r = a.order[s.fast_base[fc] + k];
    const DpRegionInfo ri = a.rinfo[r];
    const u32 tc = ri.counts & 0xFFu, qc = ri.counts >> 8;
    u32 slot_pos[2 * AVK_FAST_MAXV];
    for (u32 q = 0; q < 2 * AVK_FAST_MAXV; ++q) slot_pos[q] = 0;
    for (u32 sidx = 0; sidx < 2 * AVK_FAST_MAXV; ++sidx) {
        const u32 side = sidx / AVK_FAST_MAXV, j = sidx % AVK_FAST_MAXV;
        if (j >= maxv) continue;
        u32 *V = T + (AVK_FAST_HDR + 4 * (side * maxv + j)) * 64;
        if (j >= (side ? qc : tc)) {
            V[0] = V[64] = V[128] = V[192] = 0;
            continue;
        }
        avk_u4 sl = *(const avk_u4 *)&a.slots[(u64)(side ? ri.q_first : ri.t_first) + j]; /* written by dp_region */
        if (a.in.owned) { /* a form with explicit offsets: two regions with different starts may share this call, and the slot holds the position relative to whichever
                             of them wrote last — this region's own comes from the caller's arrays (the packed forms own their calls by construction) */
            const u64 v = (u64)(side ? ri.q_first : ri.t_first) + j;
            sl.x = (sl.x & ~0xFFu) | ((u32)(a.in.pos_of(v, a.in.start_of(r)) - a.in.start_of(r)) & 0xFFu);
        }
        slot_pos[sidx] = sl.x & 0xFFu;
        V[0] = sl.x, V[64] = sl.y, V[128] = sl.z, V[192] = sl.w;
    }
    /* order_variants (query_optimizer.rs:372-381): stable merge by position, truth first on ties; bit d = depth d takes a query call */
    u32 ord = 0, i = 0, j = 0, d = 0;
    while (i < tc || j < qc) {
        const bool take_t = j >= qc || (i < tc && slot_pos[i] <= slot_pos[AVK_FAST_MAXV + j]);
        if (take_t) i++;
        else j++, ord |= 1u << d;
        d++;
    }
    T[0] = (u32)(ri.ref_off >> 4);
    T[64] = (u32)(ri.ref_off & 15u) | (ri.len << 4) | (tc << 12) | (qc << 14) | (ord << 16);
    T[128] = a.v_off[r];
    T[192] = r;
}

/* ---- region records and blobs ------------------------------------------------------------------------------------------------- */
static const u8 DP_SUP_TYPES[8] = {AVK_VT_SNV, AVK_VT_INSERTION, AVK_VT_DELETION, AVK_VT_INDEL, AVK_VT_TR_CONTRACTION, AVK_VT_TR_EXPANSION, AVK_VT_SV_DELETION, AVK_VT_SV_INSERTION};
AVK_DEV u32 dp_sup_index(u32 vt) { /* index into AVK_SUP_TYPES or 8 */
    return vt == AVK_VT_SNV ? 0u : (vt == AVK_VT_INSERTION ? 1u : (vt == AVK_VT_DELETION ? 2u : (vt == AVK_VT_INDEL ? 3u : (vt == AVK_VT_TR_CONTRACTION ? 4u : (vt == AVK_VT_TR_EXPANSION ? 5u : (vt == AVK_VT_SV_DELETION ? 6u : (vt == AVK_VT_SV_INSERTION ? 7u : 8u)))))));
}
AVK_DEV AvkDevRegion dp_record_of(const DpArgs &a, u32 r, const DpRegionInfo &ri) {
    AvkDevRegion dr;
    dr.ref_off = ri.ref_off;
    dr.len = ri.len;
    dr.v_off = a.v_off[r];
    dr.t_cnt = a.in.t_cnt_of(r);
    dr.q_cnt = a.in.q_cnt_of(r);
    dr.pre_status = ri.pre_status;
    dr.seq_stride = ri.seq_stride;
    dr.seq_off = a.seq_off[r];
    dr.blob_off = ri.blob_bytes ? a.blob_off8[r] : 0u;
    dr.blob_bytes = ri.blob_bytes;
    dr.alle_bytes = ri.alle_bytes;
    dr.grow = ri.grow;
    dr.orig = r;
    dr.ed_bound = ri.ed_bound;
    return dr;
}
/* one lane writes record k of the work order and the region's blob (regions of at most DP_SMALL_N calls; larger ones go to big_list) */
AVK_DEV void dp_region_record(const DpArgs &a, u64 k) {
    if (k >= a.in.n_regions) return;
    const DpIn &in = a.in;
    const u32 r = a.order[k];
    const DpRegionInfo ri = a.rinfo[r];
    const AvkDevRegion dr = dp_record_of(a, r, ri);
    a.regions[k] = dr;
    if (!ri.blob_bytes) return;
    const u32 tc = dr.t_cnt, qc = dr.q_cnt, N = tc + qc;
    if (N > DP_SMALL_N) {
        const u32 slot = avk_atomic_add_u32_global(&a.st->n_big, 1u);
        a.big_list[slot] = (u32)k;
        return;
    }
    const u64 vb = ((u64)N * sizeof(AvkBlobVar) + 15) & ~15ull, ab = ((u64)ri.alle_bytes + 15) & ~15ull, ob = (u64)N * sizeof(AvkOrdVar);
    u8 *base = (u8 *)(a.blob + 2ull * dr.blob_off);
    AvkBlobVar *bv = (AvkBlobVar *)base;
    u8 *ba = base + vb;
    AvkOrdVar *bo = (AvkOrdVar *)(base + vb + ab);
    u32 *bc = (u32 *)(base + vb + ab + ob);
    { /* padding between the sections stays zero (the host packer clears the whole blob first) */
        for (u64 x = (u64)N * sizeof(AvkBlobVar); x < vb; ++x) base[x] = 0;
        for (u64 x = ri.alle_bytes; x < ab; ++x) ba[x] = 0;
    }
    const u64 start = in.start_of(r), toff_r = in.t_off_of(r), qoff_r = in.q_off_of(r);
    u32 run = 0, counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 i = 0; i < N; ++i) {
        const u64 v = i < tc ? toff_r + i : qoff_r + (i - tc);
        const u32 l0 = in.a0_len_of(v), l1 = in.a1_len_of(v);
        const u32 vt = in.type_of(v), zy = in.zyg_of(v);
        AvkBlobVar b;
        b.rel_pos = (u32)(in.pos_of(v, start) - start);
        b.a0_len = l0;
        b.a1_len = l1;
        b.a_off = run;
        b.raw_space = in.raw_of(v, l0, l1);
        b.alt_ed = a.vinfo[v].alt_ed;
        b.type_zyg = vt | (zy << 8);
        bv[i] = b;
        const u8 *s0 = in.alleles + in.a0_off_of(v), *s1 = in.alleles + in.a1_off_of(v);
        for (u32 x = 0; x < l0; ++x) ba[run + x] = s0[x];
        for (u32 x = 0; x < l1; ++x) ba[run + l0 + x] = s1[x];
        run += l0 + l1;
        const u32 si = dp_sup_index(vt);
        if (si < 8) counts[si] += i < tc ? 1u : 0x10000u;
    }
    /* order_variants: stable by position over [truth.., query..] */
    u32 i = 0, j = tc, o = 0;
    while (i < tc || j < N) {
        const u32 k2 = (j >= N || (i < tc && bv[i].rel_pos <= bv[j].rel_pos)) ? i++ : j++;
        const AvkBlobVar b = bv[k2];
        AvkOrdVar ov;
        ov.rel_pos = b.rel_pos, ov.a0_len = b.a0_len, ov.a1_len = b.a1_len, ov.a_off = b.a_off, ov.alt_ed = b.alt_ed, ov.type_zyg = b.type_zyg;
        ov.sync = dr.len;
        ov.vi = k2;
        bo[o++] = ov;
    }
    for (u32 o2 = 0; o2 + 1 < N; ++o2) bo[o2].sync = bo[o2 + 1].rel_pos;
    for (int t = 0; t < 8; ++t) bc[t] = counts[t];
}

/* one WAVE writes the blob of a region with many calls: lanes take calls; offsets by wave-wide prefix sums; the search order by rank
 * (position of a truth call = its index + the query calls before it, of a query call = its index + the truth calls not after it) */
AVK_DEV void dp_region_record_wave(const DpArgs &a, u32 item) {
    const DpIn &in = a.in;
    const u32 lane = (u32)wv_lane();
    const u32 k = a.big_list[item];
    const u32 r = a.order[k];
    const DpRegionInfo ri = a.rinfo[r];
    const u32 tc = in.t_cnt_of(r), qc = in.q_cnt_of(r), N = tc + qc;
    const u64 toff = in.t_off_of(r), qoff = in.q_off_of(r), start = in.start_of(r);
    const u32 blob_off = a.blob_off8[r];
    const u64 vb = ((u64)N * sizeof(AvkBlobVar) + 15) & ~15ull, ab = ((u64)ri.alle_bytes + 15) & ~15ull, ob = (u64)N * sizeof(AvkOrdVar);
    u8 *base = (u8 *)(a.blob + 2ull * blob_off);
    AvkBlobVar *bv = (AvkBlobVar *)base;
    u8 *ba = base + vb;
    AvkOrdVar *bo = (AvkOrdVar *)(base + vb + ab);
    u32 *bc = (u32 *)(base + vb + ab + ob);
    if (lane == 0) {
        for (u64 x = (u64)N * sizeof(AvkBlobVar); x < vb; ++x) base[x] = 0;
        for (u64 x = ri.alle_bytes; x < ab; ++x) ba[x] = 0;
    }
    u32 carry = 0;
    u32 cnt_lo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 i0 = 0; i0 < N; i0 += 64) {
        const u32 i = i0 + lane;
        const bool on = i < N;
        const u64 v = on ? (i < tc ? toff + i : qoff + (i - tc)) : 0;
        const u32 l0 = on ? in.a0_len_of(v) : 0u, l1 = on ? in.a1_len_of(v) : 0u;
        /* exclusive prefix sum of l0 + l1 over the 64 lanes */
        u32 x = l0 + l1, incl = x;
        for (u32 d = 1; d < 64; d <<= 1) {
            const u32 y = wv_shfl(incl, (int)(lane >= d ? lane - d : 0));
            if (lane >= d) incl += y;
        }
        const u32 a_off = carry + incl - x;
        carry += wv_shfl(incl, 63);
        if (on) {
            const u32 vt = in.type_of(v), zy = in.zyg_of(v);
            AvkBlobVar b;
            b.rel_pos = (u32)(in.pos_of(v, start) - start);
            b.a0_len = l0;
            b.a1_len = l1;
            b.a_off = a_off;
            b.raw_space = in.raw_of(v, l0, l1);
            b.alt_ed = a.vinfo[v].alt_ed;
            b.type_zyg = vt | (zy << 8);
            bv[i] = b;
            const u8 *s0 = in.alleles + in.a0_off_of(v), *s1 = in.alleles + in.a1_off_of(v);
            for (u32 q = 0; q < l0; ++q) ba[a_off + q] = s0[q];
            for (u32 q = 0; q < l1; ++q) ba[a_off + l0 + q] = s1[q];
            const u32 si = dp_sup_index(vt);
            if (si < 8) cnt_lo[si] += i < tc ? 1u : 0x10000u;
        }
    }
    for (int t = 0; t < 8; ++t) {
        const u32 s = wv_sum_u32(cnt_lo[t]); /* at most 60000 calls: the two 16-bit halves cannot carry into each other */
        if (lane == 0) bc[t] = s;
    }
    wv_sync(); /* the records above are read below by other lanes */
    for (u32 i0 = 0; i0 < N; i0 += 64) {
        const u32 i = i0 + lane;
        if (i < N) {
            const AvkBlobVar b = bv[i];
            u32 rank;
            if (i < tc) { /* query calls strictly before this position */
                u32 lo = 0, hi = qc;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (bv[tc + mid].rel_pos < b.rel_pos) lo = mid + 1;
                    else hi = mid;
                }
                rank = i + lo;
            } else { /* truth calls at or before this position */
                u32 lo = 0, hi = tc;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (bv[mid].rel_pos <= b.rel_pos) lo = mid + 1;
                    else hi = mid;
                }
                rank = (i - tc) + lo;
            }
            AvkOrdVar ov;
            ov.rel_pos = b.rel_pos, ov.a0_len = b.a0_len, ov.a1_len = b.a1_len, ov.a_off = b.a_off, ov.alt_ed = b.alt_ed, ov.type_zyg = b.type_zyg;
            ov.sync = ri.len;
            ov.vi = i;
            bo[rank] = ov;
        }
    }
    wv_sync();
    for (u32 o0 = 0; o0 + 1 < N; o0 += 64) {
        const u32 o = o0 + lane;
        if (o + 1 < N) bo[o].sync = bo[o + 1].rel_pos;
    }
}

/* ---- dp_unpack: the results in the caller's layout (avk_result_batch), one lane per region --------------------------------------- */
struct DpOut {
    const u32 *region_out; /* [n][4] */
    const u32 *var_out;    /* [total_v] */
    const u32 *v_off;
    const u64 *t_off, *q_off;
    const u32 *t_cnt, *q_cnt;
    u64 n_regions, n_variants;
    u64 v_lo; /* the per-call arrays below start at call v_lo of the caller's arrays */
    int32_t *status;
    u32 *ed_h1, *ed_h2, *n_optima;
    uint16_t *type_present;
    u8 *var_expected, *var_observed, *var_class, *var_zyg; /* any of them may be NULL */
    u32 mode; /* 1: the pair form — no per-call outputs */
    u64 *region_packed; /* the packed form (avk_result_batch::region_packed / var_packed), or NULL; then status may be NULL too */
    u8 *var_packed;
    /* the packed form of the BASEPAIR groups (avk_result_batch::bp_packed), or NULL: from the kernels' groups (bp_off_dev / bp_dev, 4 counters each) one word per
     * region; the groups of a region that needs more are spilled behind an atomic counter */
    const u32 *bp_off_dev, *bp_dev;
    u32 *bp_packed, *bp_spill, *bp_spill_count;
    /* a batch read from its packed source (DpIn::pk_*): t_off / q_off / t_cnt / q_cnt above are NULL, the counts and the running sum of the calls stand in */
    const u64 *pk_voff;
    const u8 *pk_tc, *pk_qc;
};
/* the packed BASEPAIR word of region r, or — the region needs its groups spilled — how many groups that is (then `word` is not set) */
AVK_DEV u32 dp_unpack_bp_need(const DpOut &o, u64 r, u32 &word) {
    word = 0;
    if (r >= o.n_regions || !o.bp_packed) return 0;
    const u32 lo = o.bp_off_dev[r], hi = o.bp_off_dev[r + 1];
    if (o.region_out[4 * r] != 0 || hi <= lo) return 0;
    const avk_u4 j = *(const avk_u4 *)(o.bp_dev + 4 * (u64)lo);
    bool simple = (j.x | j.y | j.z | j.w) < 128u;
    for (u32 k = lo + 1; k < hi && simple; ++k) {
        const avk_u4 g = *(const avk_u4 *)(o.bp_dev + 4 * (u64)k);
        simple = g.x == j.x && g.y == j.y && g.z == j.z && g.w == j.w;
    }
    if (simple) {
        word = j.x | (j.y << 7) | (j.z << 14) | (j.w << 21);
        return 0;
    }
    return hi - lo;
}
/* `need` / `word` from dp_unpack_bp_need; `at` = where the region's spilled groups go (the kernel hands the places out a workgroup at a time: one atomic per 256
 * regions instead of one per wave — 56,000 returning atomics on one word were 0.4 ms of a whole-genome call); at == 0xFFFFFFFF: taken here, with an atomic of its own */
AVK_DEV void dp_unpack(const DpOut &o, u64 r, u32 need, u32 word, u32 at) {
    if (r >= o.n_regions) return;
    const avk_u4 w = *(const avk_u4 *)(o.region_out + 4 * r);
    if (o.bp_packed) {
        if (need) {
            const u32 lo = o.bp_off_dev[r];
            if (at == 0xFFFFFFFFu) at = avk_atomic_add_u32_global(o.bp_spill_count, need);
            for (u32 k = 0; k < need; ++k) *(avk_u4 *)(o.bp_spill + 4 * (u64)(at + k)) = *(const avk_u4 *)(o.bp_dev + 4 * (u64)(lo + k));
            word = 0x80000000u | at;
        }
        o.bp_packed[r] = word;
    }
    if (o.status) o.status[r] = (int32_t)w.x;
    if (o.ed_h1) o.ed_h1[r] = w.y;
    if (o.ed_h2) o.ed_h2[r] = w.z;
    if (o.n_optima) o.n_optima[r] = w.w & 0xFFFFu;
    if (o.type_present) o.type_present[r] = (uint16_t)(w.w >> 16);
    if (o.region_packed) { /* avk_rp_make of the public header */
        const u64 e1 = w.y < AVK_RP_ED_MAX ? w.y : AVK_RP_ED_MAX, e2 = w.z < AVK_RP_ED_MAX ? w.z : AVK_RP_ED_MAX;
        const u64 filtered = ((w.w >> 16) & AVK_FILTERED_TYPE_MASK) == AVK_FILTERED_TYPE_MASK ? 1u : 0u;
        o.region_packed[r] = (u64)(w.x & 0x7Fu) | filtered << 7 | (u64)(w.w & 0xFFFFu) << 8 | e1 << 24 | e2 << 44;
    }
    if (o.mode != 0 || !(o.var_expected || o.var_observed || o.var_class || o.var_zyg || o.var_packed)) return;
    const u32 tc = o.pk_voff ? (u32)o.pk_tc[r] : o.t_cnt[r], qc = o.pk_voff ? (u32)o.pk_qc[r] : o.q_cnt[r];
    const u64 toff = o.pk_voff ? o.pk_voff[r] : o.t_off[r], qoff = o.pk_voff ? toff + tc : o.q_off[r];
    if (toff > o.n_variants || (u64)tc > o.n_variants - toff || qoff > o.n_variants || (u64)qc > o.n_variants - qoff) return;
    const u32 *vw = o.var_out + o.v_off[r];
    for (u32 k = 0; k < tc + qc; ++k) {
        const u64 hv = (k < tc ? toff + k : qoff + (k - tc)) - o.v_lo;
        const u32 x = vw[k];
        if (o.var_expected) o.var_expected[hv] = (u8)(x & 0xFF);
        if (o.var_observed) o.var_observed[hv] = (u8)((x >> 8) & 0xFF);
        if (o.var_class) o.var_class[hv] = (u8)((x >> 16) & 0xFF);
        if (o.var_zyg) o.var_zyg[hv] = (u8)(x >> 24);
        if (o.var_packed) o.var_packed[hv] = (u8)((x & 3u) | ((x >> 8) & 3u) << 2 | ((x >> 24) & 7u) << 4);
    }
}

AVK_DEV void dp_unpack(const DpOut &o, u64 r) { /* one region by itself (the emulator, the tests) */
    u32 word = 0;
    const u32 need = dp_unpack_bp_need(o, r, word);
    dp_unpack(o, r, need, word, 0xFFFFFFFFu);
}

} // namespace dp
} // namespace avk
#endif
