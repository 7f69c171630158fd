/*
 * aardvark_amd.h — C-ABI of the MI355X (gfx950) compare hot path.
 *
 * This is the drop-in boundary for the per-region solver of `aardvark compare`:
 * one call replaces the reference's rayon loop
 *     all_regions.into_par_iter().map(|r| solve_compare_region(&r, &genome, cfg, strat))
 * (reference src/main.rs:251-268, src/waffle_solver.rs:122-124) with a batched launch
 * of hand-written HIP kernels.  The reference has no FFI of its own; the flat
 * structures below carry exactly the fields of `CompareRegion`
 * (src/data_types/compare_region.rs:13-26), `Variant` (src/data_types/variants.rs:73-91),
 * `CompareConfig` (src/waffle_solver.rs:94-103) and `CompareBenchmark`
 * (src/data_types/compare_benchmark.rs:9-33).
 *
 * Conventions
 *  - plain pointers + sizes, no C++/torch types; the caller owns every buffer;
 *    nothing is retained after a call returns except what a context owns
 *    (uploaded reference, uploaded batches).
 *  - every function returns 0 on success or a negative AVK_E_* infrastructure code;
 *    per-region solver outcomes are reported in `status[]` (AVK_ST_*), mirroring the
 *    reference's `anyhow::Result` per region (src/main.rs:255-265): a failed region
 *    produces no metrics and the batch continues.
 *  - coordinates are 0-based half-open (src/data_types/coordinates.rs:6-13).
 */
#ifndef AARDVARK_AMD_H
#define AARDVARK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (ordinals identical to the reference's Rust enums) ------------------ */

/* VariantType, src/data_types/variants.rs:6-31 */
enum {
    AVK_VT_SNV = 0, AVK_VT_INSERTION = 1, AVK_VT_DELETION = 2, AVK_VT_INDEL = 3,
    AVK_VT_SV_INSERTION = 4, AVK_VT_SV_DELETION = 5, AVK_VT_SV_DUPLICATION = 6,
    AVK_VT_SV_INVERSION = 7, AVK_VT_SV_BREAKEND = 8, AVK_VT_TR_CONTRACTION = 9,
    AVK_VT_TR_EXPANSION = 10, AVK_VT_UNKNOWN = 11, AVK_N_VARIANT_TYPES = 12
};

/* PhasedZygosity, src/data_types/phase_enums.rs:33-46 */
enum {
    AVK_ZYG_UNKNOWN = 0, AVK_ZYG_HOM_REF = 1, AVK_ZYG_UNPHASED_HET = 2,
    AVK_ZYG_PHASED_HET01 = 3, AVK_ZYG_PHASED_HET10 = 4, AVK_ZYG_HOM_ALT = 5
};

/* Classification, src/data_types/variant_metrics.rs:12-22 */
enum { AVK_CLASS_UNKNOWN = 0, AVK_CLASS_TP = 1, AVK_CLASS_FN = 2, AVK_CLASS_FP = 3 };

/* Metric block geometry.  A region's GroupTypeMetrics (src/data_types/grouped_metrics.rs:32-37)
 * is 13 groups (group 0 = joint, group 1+t = VariantType t) of 22 counters
 * (GroupMetrics, grouped_metrics.rs:150-161 + summary_metrics.rs:6-15,84-92). */
#define AVK_N_GROUPS 13
#define AVK_N_FIELDS 22
enum {
    AVK_F_GT_TRUTH_TP = 0, AVK_F_GT_TRUTH_FN, AVK_F_GT_QUERY_TP, AVK_F_GT_QUERY_FP,
    AVK_F_GT_TRUTH_FN_GT, AVK_F_GT_QUERY_FP_GT,
    AVK_F_HAP_TRUTH_TP, AVK_F_HAP_TRUTH_FN, AVK_F_HAP_QUERY_TP, AVK_F_HAP_QUERY_FP,
    AVK_F_WHAP_TRUTH_TP, AVK_F_WHAP_TRUTH_FN, AVK_F_WHAP_QUERY_TP, AVK_F_WHAP_QUERY_FP,
    AVK_F_BP_TRUTH_TP, AVK_F_BP_TRUTH_FN, AVK_F_BP_QUERY_TP, AVK_F_BP_QUERY_FP,
    AVK_F_RBP_TRUTH_TP, AVK_F_RBP_TRUTH_FN, AVK_F_RBP_QUERY_TP, AVK_F_RBP_QUERY_FP
};
/* tally block = AVK_N_GROUPS*AVK_N_FIELDS sums, then solved_blocks, error_blocks
 * (src/writers/summary.rs:146-163) */
#define AVK_TALLY_LEN (AVK_N_GROUPS * AVK_N_FIELDS + 2)
#define AVK_TALLY_SOLVED (AVK_N_GROUPS * AVK_N_FIELDS)
#define AVK_TALLY_ERRORS (AVK_N_GROUPS * AVK_N_FIELDS + 1)

/* per-region status: 0 = Ok(CompareBenchmark); >0 = the reference would return Err
 * or panic for this region (class of the error) */
enum {
    AVK_ST_OK = 0,
    AVK_ST_BRANCH_FACTOR = 2,   /* "max_branch_factor must be greater than 0", query_optimizer.rs:177 */
    AVK_ST_NO_RESULTS = 3,      /* "no results found", query_optimizer.rs:331 */
    AVK_ST_NO_GT_RESULT = 4,    /* "No result found for problem", exact_gt_optimizer.rs:347 */
    AVK_ST_UNKNOWN_ALLELE = 5,  /* Allele::Unknown reached an optimizer, exact_gt_optimizer.rs:256 */
    AVK_ST_BAD_ZYGOSITY = 6,    /* assert_eq!(zyg, HomozygousAlternate) panics, query_optimizer.rs:315 */
    AVK_ST_VARIANT_METRICS = 7, /* VariantMetrics::new(.., 0, 0), variant_metrics.rs:57 */
    AVK_ST_TRUTH_FP = 8,        /* assert!(exp >= obs), waffle_solver.rs:322 */
    AVK_ST_RECORD_BP = 9,       /* "Truth/Query TP is less than basepair TP", waffle_solver.rs:492-493 */
    AVK_ST_SEQ_MISMATCH = 10,   /* assert_eq!(regenerated, optimizer sequence), waffle_solver.rs:365-367 */
    AVK_ST_AUTOFAIL_OOB = 11,   /* all_variant_order[auto_fail_index] out of bounds, exact_gt_optimizer.rs:312 */
    AVK_ST_INVALID_INPUT = 20,  /* rejected by host validation (window outside contig, variant outside
                                   window, unsorted variants, empty allele): the reference panics or is undefined */
    AVK_ST_CAPACITY = 21        /* device workspace exhausted at the largest tier AND in the library's own retries with slices of
                                   1, 4 and 16 GB (avk_results_download, context option "capacity_retry", default 1).  The retries repair
                                   what avk_results_download hands to the caller (per-region and per-call arrays, out->tally) and the
                                   batch's region records and metric blocks on the device, so that avk_label_tallies AFTER the download
                                   counts the repaired regions as the totals do; tally_dev of avk_compare_resident was written before
                                   the download and still counts a retried region as failed */
};

/* infrastructure errors (function return values) */
enum {
    AVK_E_OK = 0, AVK_E_ARG = -1, AVK_E_HIP = -2, AVK_E_OOM = -3, AVK_E_STATE = -4
};

/* ---- inputs ---------------------------------------------------------------------- */

/* A batch of CompareRegions in structure-of-arrays form.  Region r owns truth variants
 * [t_off[r], t_off[r]+t_cnt[r]) and query variants [q_off[r], q_off[r]+q_cnt[r]) of the
 * variant arrays; variant v's alleles are allele_bytes[a0_off[v] .. +a0_len[v]) (allele0 =
 * VCF REF after trimming) and [a1_off[v] .. +a1_len[v]) (allele1 = ALT). */
typedef struct avk_region_batch {
    uint64_t n_regions;
    const uint64_t *region_id;   /* CompareRegion::region_id */
    const uint32_t *contig_idx;  /* index into the uploaded reference (Coordinates::chrom) */
    const uint64_t *start;       /* Coordinates::start */
    const uint64_t *end;         /* Coordinates::end (exclusive) */
    const uint64_t *t_off;
    const uint32_t *t_cnt;
    const uint64_t *q_off;
    const uint32_t *q_cnt;

    uint64_t n_variants;
    const uint64_t *var_pos;       /* Variant::position (0-based, contig coordinates) */
    const uint8_t  *var_type;      /* AVK_VT_* */
    const uint8_t  *var_zyg;       /* AVK_ZYG_* (input zygosity of the call) */
    const uint32_t *var_raw_space; /* Variant::raw_allele_space */
    const uint64_t *a0_off;
    const uint32_t *a0_len;
    const uint64_t *a1_off;
    const uint32_t *a1_len;

    const uint8_t *allele_bytes;
    uint64_t allele_bytes_len;
} avk_region_batch;

/* The same batch in the library's COMPACT form: 20 bytes per region and 17 (21 with raw_allele_space) per call instead of 52 and 38 — the form to
 * build when the batch crosses PCIe once per call (a whole genome: 0.23 GB instead of 0.48 GB; the host boundary is bound by that copy).  Constraints
 * that make the narrow fields possible: contigs shorter than 4 Gbp, the calls of a region adjacent in the call arrays (truth calls at
 * [v_off, v_off + t_cnt), query calls right behind them), allele1 of a call right behind its allele0 in allele_bytes, fewer than 2^32 calls and
 * allele bytes, at most 65535 calls per region and side.  The library widens it on the device (one kernel); results are indexed like the arrays here. */
typedef struct avk_compact_batch {
    uint64_t n_regions;
    const uint32_t *contig_idx;    /* [n_regions] may be NULL (contig 0) */
    const uint32_t *start;         /* [n_regions] Coordinates::start */
    const uint32_t *len;           /* [n_regions] end - start */
    const uint32_t *v_off;         /* [n_regions] first call of the region */
    const uint16_t *t_cnt, *q_cnt; /* [n_regions] */
    uint64_t n_variants;
    const uint32_t *var_pos;       /* [n_variants] Variant::position */
    const uint8_t  *var_type_zyg;  /* [n_variants] AVK_VT_* | AVK_ZYG_* << 4 */
    const uint32_t *a_off;         /* [n_variants] allele0 at allele_bytes[a_off ..), allele1 right behind it */
    const uint32_t *a0_len, *a1_len;
    const uint32_t *var_raw_space; /* [n_variants] may be NULL (= the longer allele) */
    const uint8_t  *allele_bytes;
    uint64_t allele_bytes_len;
} avk_compact_batch;

/* The same batch in the PACKED form (round 3): 10 bytes per region and 5 per call plus the allele bytes — what is left when every offset the other forms
 * carry is implied by order: the calls of region r follow those of region r - 1 (truth calls, then query calls), the alleles of call v follow those of
 * call v - 1 (allele0, then allele1), call positions are relative to their region's start.  The library computes the offsets on the device (two prefix
 * sums) and writes the wide arrays there.  A whole genome: 94 MB over PCIe instead of 227 (compact) or 479 (wide).  Constraints: windows shorter than
 * 65,536 bases, at most 255 calls per region and side, alleles of at most 255 bases, fewer than 2^32 calls and allele bytes, contigs shorter than 4 Gbp,
 * at most 65,535 contigs.  Results are indexed like the arrays here. */
typedef struct avk_packed_batch {
    uint64_t n_regions;
    const uint16_t *contig_idx;    /* [n_regions] may be NULL (contig 0) */
    const uint32_t *start;         /* [n_regions] Coordinates::start */
    const uint16_t *len;           /* [n_regions] end - start */
    const uint8_t  *t_cnt, *q_cnt; /* [n_regions] */
    uint64_t n_variants;           /* = sum of t_cnt + q_cnt */
    const uint16_t *var_rel_pos;   /* [n_variants] Variant::position - the region's start */
    const uint8_t  *var_type_zyg;  /* [n_variants] AVK_VT_* | AVK_ZYG_* << 4 */
    const uint8_t  *a0_len, *a1_len; /* [n_variants] */
    const uint32_t *var_raw_space; /* [n_variants] may be NULL (= the longer allele) */
    const uint8_t  *allele_bytes;  /* allele0 then allele1 of call 0, of call 1, ... */
    uint64_t allele_bytes_len;     /* = sum of a0_len + a1_len */
} avk_packed_batch;

/* CompareConfig, src/waffle_solver.rs:94-115 */
typedef struct avk_compare_config {
    uint32_t max_branch_factor;     /* default 50 */
    uint32_t enable_sequences;      /* fill seq_* outputs (debug region_sequences.tsv.gz) */
    uint32_t enable_exact_shortcut; /* hidden --enable-exact-shortcut */
} avk_compare_config;

/* ---- outputs --------------------------------------------------------------------- */

/* Caller-allocated result arrays, positionally aligned with the input batch.
 * Any pointer except `status` may be NULL to skip that output (and `status` too when region_packed is given). */
typedef struct avk_result_batch {
    int32_t  *status;        /* [n_regions] AVK_ST_* */
    uint32_t *ed_h1;         /* [n_regions] CompareBenchmark::bm_edit_distance_h1 */
    uint32_t *ed_h2;         /* [n_regions] */
    uint32_t *n_optima;      /* [n_regions] tied optima returned by optimize_sequences */
    uint16_t *type_present;  /* [n_regions] bit t: VariantType t has an entry in variant_metrics */
    uint32_t *group_metrics; /* [n_regions][AVK_N_GROUPS][AVK_N_FIELDS] full GroupTypeMetrics */

    /* per variant, same indexing as the input variant arrays; query entries are already
     * toggled the way CompareBenchmark::add_swap_benchmark stores them
     * (compare_benchmark.rs:109-123): these are the EA / OA / BD values of the output VCFs */
    uint8_t  *var_expected;  /* [n_variants] */
    uint8_t  *var_observed;  /* [n_variants] */
    uint8_t  *var_class;     /* [n_variants] AVK_CLASS_* */
    uint8_t  *var_zyg;       /* [n_variants] zygosity resolved by the phasing search (GT of output VCFs) */

    /* SequenceBundle (compare_benchmark.rs:166-185) when cfg.enable_sequences:
     * region r, sequence k (0 ref, 1 truth1, 2 truth2, 3 query1, 4 query2) is
     * seq_bytes[seq_off[r] + k*seq_stride[r] .. + seq_len[5*r+k]).  seq_off/seq_stride are
     * INPUTS sized with avk_seq_stride(). */
    uint8_t  *seq_bytes;
    const uint64_t *seq_off;    /* [n_regions] */
    const uint32_t *seq_stride; /* [n_regions] */
    uint32_t *seq_len;          /* [n_regions][5] */

    uint64_t *tally;         /* [AVK_TALLY_LEN] sums over the Ok regions of this batch */

    /* Compact per-region BASEPAIR groups (optional; both or neither): what is left of a region's GroupTypeMetrics once the per-call decisions above are known.
     * Region r owns groups [bp_off[r], bp_off[r + 1]) of bp_groups: the joint group first, then one per VariantType that occurs among the region's
     * calls, in type order; a group is 4 counters: BASEPAIR truth_tp, truth_fn, query_tp, query_fp (summary_metrics.rs:6-15; the doubled values of
     * waffle_solver.rs:335-449).  16 bytes x (1 + call types) per region instead of the 1144-byte block of group_metrics; regions that fail validation own no
     * group, regions that fail in the solver own zeros.  avk_group_metrics_from_compact rebuilds the full 13 x 22 block of a region from these and the
     * per-call outputs. */
    uint32_t *bp_off;        /* [n_regions + 1] */
    uint32_t *bp_groups;     /* [capacity][4] with capacity >= n_regions + n_variants (a region has at most 1 + its number of calls groups) */

    /* The packed result form (optional, each on its own): everything the arrays above say per region in 8 bytes and per call in 1 byte — a whole-genome
     * job's results cross PCIe as 37 MB instead of 96.  A caller that hands in region_packed may leave `status` NULL; one that hands in var_packed needs none
     * of var_expected / var_observed / var_class / var_zyg.  Read them with the avk_rp_* / avk_vp_* accessors below (what the FFI side would do while it
     * builds its CompareBenchmark values) or expand them into the arrays above with avk_results_expand.
     *   region_packed[r] = status (bits 0-6) | "the 8 filtered types have map entries" (bit 7) | n_optima (bits 8-23) | ed_h1 (bits 24-43) | ed_h2 (bits 44-63);
     *                      an edit distance of 2^20 - 1 or more reads as AVK_RP_ED_MAX: ask for ed_h1 / ed_h2 as well if regions can be that far from the
     *                      reference (none of a genome's is: the search gives up on such windows long before, AVK_ST_CAPACITY)
     *   var_packed[v]    = expected (bits 0-1) | observed (bits 2-3) | resolved zygosity AVK_ZYG_* (bits 4-6); the class follows from the two counts and the
     *                      side the call is on (variant_metrics.rs:43-101): avk_vp_class */
    uint64_t *region_packed; /* [n_regions] */
    uint8_t  *var_packed;    /* [n_variants] */

    /* The BASEPAIR groups in their packed form (optional; hand in bp_packed, bp_spilled AND bp_groups, leave bp_off NULL): one word per region instead of
     * 4 + 16 x (1 + call types) bytes.  A region whose groups all equal its joint group — every region with calls of ONE type: a side filtered to the type it
     * consists of is the side itself (waffle_solver.rs:383-445) — and whose four counters are below 128 is that word:
     *   bp_packed[r] = truth_tp | truth_fn << 7 | query_tp << 14 | query_fp << 21                         (bit 31 clear)
     * any other region's groups (the joint one, then one per call type, as above) are SPILLED into bp_groups, in no particular order:
     *   bp_packed[r] = 0x80000000 | index of the region's first group in bp_groups
     * bp_spilled[0] = number of groups in bp_groups.  A genome's regions need 4 + ~2.5 bytes each this way instead of 36 (profiles/r05_bp_packed.txt); regions with a
     * non-zero status hold 0.  avk_group_metrics_from_compact reads either form.  Device-packed batches only (option device_pack, the default). */
    uint32_t *bp_packed;     /* [n_regions] */
    uint32_t *bp_spilled;    /* [1] */
} avk_result_batch;

#define AVK_BP_SPILL 0x80000000u
static inline int      avk_bp_is_spilled(uint32_t w) { return (w & AVK_BP_SPILL) != 0; }
static inline uint32_t avk_bp_spill_index(uint32_t w) { return w & 0x7FFFFFFFu; }
static inline uint32_t avk_bp_counter(uint32_t w, int i) { return (w >> (7 * i)) & 0x7Fu; } /* i: 0 truth_tp, 1 truth_fn, 2 query_tp, 3 query_fp */

#define AVK_RP_ED_MAX 0xFFFFFu
/* the 8 variant types add_basepair_stats filters by (waffle_solver.rs:383-445): a solved region has map entries for them whether they occur or not */
#define AVK_FILTERED_TYPE_MASK ((1u << AVK_VT_SNV) | (1u << AVK_VT_INSERTION) | (1u << AVK_VT_DELETION) | (1u << AVK_VT_INDEL) | (1u << AVK_VT_TR_CONTRACTION) | \
                                (1u << AVK_VT_TR_EXPANSION) | (1u << AVK_VT_SV_DELETION) | (1u << AVK_VT_SV_INSERTION))
static inline uint64_t avk_rp_make(uint32_t status, uint32_t ed_h1, uint32_t ed_h2, uint32_t n_optima, uint32_t type_present) {
    const uint64_t e1 = ed_h1 < AVK_RP_ED_MAX ? ed_h1 : AVK_RP_ED_MAX, e2 = ed_h2 < AVK_RP_ED_MAX ? ed_h2 : AVK_RP_ED_MAX;
    const uint64_t filtered = (type_present & AVK_FILTERED_TYPE_MASK) == AVK_FILTERED_TYPE_MASK ? 1u : 0u;
    return (uint64_t)(status & 0x7Fu) | filtered << 7 | (uint64_t)(n_optima & 0xFFFFu) << 8 | e1 << 24 | e2 << 44;
}
static inline int32_t  avk_rp_status(uint64_t w) { return (int32_t)(w & 0x7Fu); }
static inline uint32_t avk_rp_filtered_types(uint64_t w) { return (uint32_t)(w >> 7) & 1u; }
static inline uint32_t avk_rp_n_optima(uint64_t w) { return (uint32_t)(w >> 8) & 0xFFFFu; }
static inline uint32_t avk_rp_ed_h1(uint64_t w) { return (uint32_t)(w >> 24) & AVK_RP_ED_MAX; }
static inline uint32_t avk_rp_ed_h2(uint64_t w) { return (uint32_t)(w >> 44) & AVK_RP_ED_MAX; }
/* type_present of a region from its packed word and the OR of (1 << type) over its calls */
static inline uint16_t avk_rp_type_present(uint64_t w, uint32_t call_types) {
    return avk_rp_status(w) != 0 ? (uint16_t)0 : (uint16_t)(call_types | (avk_rp_filtered_types(w) ? AVK_FILTERED_TYPE_MASK : 0u));
}
static inline uint8_t avk_vp_make(uint32_t expected, uint32_t observed, uint32_t zyg) { return (uint8_t)((expected & 3u) | (observed & 3u) << 2 | (zyg & 7u) << 4); }
static inline uint8_t avk_vp_expected(uint8_t b) { return (uint8_t)(b & 3u); }
static inline uint8_t avk_vp_observed(uint8_t b) { return (uint8_t)((b >> 2) & 3u); }
static inline uint8_t avk_vp_zyg(uint8_t b) { return (uint8_t)((b >> 4) & 7u); }
/* TP when the counts agree; otherwise a truth call is a FN and a query call (stored toggled) a FP; a call of a failed region has no class */
static inline uint8_t avk_vp_class(uint8_t b, int is_query) {
    if ((b & 15u) == 0) return AVK_CLASS_UNKNOWN;
    return avk_vp_expected(b) == avk_vp_observed(b) ? AVK_CLASS_TP : (is_query ? AVK_CLASS_FP : AVK_CLASS_FN);
}

/* ---- context --------------------------------------------------------------------- */

typedef struct avk_ctx avk_ctx;          /* one per GPU / per host thread */
typedef struct avk_dev_batch avk_dev_batch; /* a region batch resident in HBM */

int  avk_ctx_create(int device_id, avk_ctx **out);
void avk_ctx_destroy(avk_ctx *ctx);
/* last error text of this context (or of the failed create when ctx == NULL) */
const char *avk_last_error(const avk_ctx *ctx);
/* run every launch of this context on an existing hipStream_t (e.g. torch's current stream).  The call BLOCKS until the stream the context used so far is idle
 * (its buffer pool hands memory out in the order of one stream); when that one was the caller's too it should still be alive — if it has been destroyed the
 * call waits for the whole device instead and still switches. */
int  avk_ctx_set_stream(avk_ctx *ctx, void *hip_stream);
/* Context options, forty of them (set them before avk_batch_upload: the work plan of a batch is made at upload).  None of them changes a result: the parity tests run under
 * random sets of them (tests/test_gpu_random_schedules.py).  Round 6 removed 27 names whose A/B comparisons were settled (profiles/r06_options.txt lists them with
 * the value each is now fixed at); what is left is what selects a code path, sizes a workspace, or is needed to reproduce a measurement.
 *   workspace tiers  "lds_bytes_per_wave" (10240), "lds_ed_cap" (48): the small LDS slice of the wave-per-region kernels; "lds2_bytes_per_wave" (40960), "lds2_ed_cap" (48):
 *                    the large one; "ws_bytes_per_wave" (1 MiB): the per-wave HBM slice, "adaptive_ws" (1: sized from the batch's predicted needs), "ws_budget_bytes";
 *                    "big_ws_bytes" (64 MiB): the shared big slices; "hbm_ed_cap"; "waves_per_cu"; 0 bytes disables a tier; "capacity_retry" (1: a region no tier could hold is
 *                    solved again by the library in larger slices)
 *   scheduling       "solo_min_variants" (5: regions with at least this many calls go to the solo launches), "class_c_nodes_x2" (12: nodes per call a search is priced at
 *                    when class C is decided), "class_c_below" (16384: a batch with lane launches and at most this many regions outside them plans those regions for the
 *                    wave-cooperative kernel — a contig, a rank's shard),
 *                    "lds_escalation" (1: in-workgroup escalation of the bulk launch), "static_pct" (75: share of a launch's work list dealt statically), "claim" (2:
 *                    regions per dynamic claim)
 *   lane kernels     "lane_kernel" (1: regions with at most three calls per side on a short window are solved one per LANE, avk_lane.inl / four lanes per region,
 *                    avk_quad.inl), "lane_quad" (1: launches of at most 16 records per wave run four lanes per region), "lane_pairs" (1: regions with the same SNV on
 *                    both sides are looked up, avk_pairs.inl), "lane_min_regions" (2048) / "lane_min_batch" (16384): smaller classes / batches stay with the
 *                    wave-per-region kernels, "lane_node_cap" (32: search nodes the three-call class makes before it hands a region over), "lane_width_one" / "_two" /
 *                    "_three" (64, 64, 16: records a wave takes at a time), "lane_head_width" (16: records per wave in the heads of the classes, the tiles of regions
 *                    with estimated edits; 0 = no head launch), "use_packed_reference" (1: the 2-bit reference; 0: bytes — every region then goes to the wave kernels)
 *   wide kernel      "wide_kernel" (1: large searches on small windows are solved one per WAVE, avk_wide.inl), "wide_lds_bytes" (16384), "wide_retry_lds_bytes" (65536)
 *   long windows     "team_long_windows" (1: a batch of large windows — no region of class C is the wide kernel's — runs the head of the class a WORKGROUP per region,
 *                    avk_region_kernel_team; 0: a wave per region; 2: a workgroup per region whose owner wave takes every job itself, a diagnostic),
 *                    "team_head_regions" (48: how many regions that head has)
 *   boundary         "device_pack" (1: batches are validated, classified, ordered and written in the kernels' layout on the device, avk_devpack.inl; 0: by host threads,
 *                    avk_pack.h — same records either way), "packed_source" (1: an avk_packed_batch is packed from the packed arrays themselves; 0: from a wide copy made
 *                    on the device first, round 5), "kernel_copies" (1: every synchronous call that copies by DMA engine times its own copies in; in every second process of a box they crawl at half the
 *                    link's rate, and from the first such call on the pinned arrays of the context's synchronous calls cross the bus by a copy kernel, every lane 16
 *                    bytes at a time from / to the pinned pages: profiles/r06_copy_engines.txt; 0: always hipMemcpyAsync; 2: always the kernel),
 *                    "split_parts" (1; 2..4: an avk_compare_packed call of pinned arrays runs as that many batches in flight,
 *                    compare_packed_split — slower than the whole call today, profiles/r06_split_call.txt)
 *   outputs          "emit_group_metrics" (0 = kernels skip the per-region 13 x 22 block; the batch tally is always produced), "emit_bp_groups" (1 = kernels write the
 *                    compact per-region BASEPAIR groups; the one-call entry points switch it on when the caller hands the arrays in), "accumulate_tally" (1 =
 *                    avk_compare_resident ADDS the batch tally to tally_dev)
 * The launches of one call run on HIP streams side by side; the HIP runtime gives a process 4 hardware queues by default and streams that share one take turns:
 * avk_ctx_create sets GPU_MAX_HW_QUEUES=24 unless the environment already has it (effective when it is the process's first HIP call). */
int  avk_ctx_set_option(avk_ctx *ctx, const char *name, int64_t value);

/* Replaces ReferenceGenome::from_fasta + get_full_chromosome (src/main.rs:94,
 * src/waffle_solver.rs:131): contigs are copied to HBM once (2-bit packed plus the raw
 * bytes needed for windows holding non-ACGT symbols) and owned by the context. */
int  avk_ref_upload(avk_ctx *ctx, uint32_t n_contigs, const uint8_t *const *seqs, const uint64_t *lens);

/* One call = solve_compare_region for every region of the batch: the caller's arrays go to HBM as they are, packing kernels, solver
 * kernels, unpacking kernel, results into the caller's arrays (no per-region work on the host). */
int  avk_compare_batch(avk_ctx *ctx, const avk_region_batch *batch,
                       const avk_compare_config *cfg, avk_result_batch *out);

/* the same for a batch in the compact form */
int  avk_compare_compact(avk_ctx *ctx, const avk_compact_batch *batch, const avk_compare_config *cfg, avk_result_batch *out);
int  avk_batch_upload_compact(avk_ctx *ctx, const avk_compact_batch *batch, avk_dev_batch **out);
/* and in the packed form (replaces the same loop, src/main.rs:251-268; the batch crosses PCIe as 94 MB per whole genome) */
int  avk_compare_packed(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out);

/* The same call in two halves, for a caller with several batches (the reference streams its regions through one rayon loop and collects at the end,
 * src/main.rs:251-268): batches in flight inside ONE context.  avk_compare_packed_submit copies the batch's arrays on the context's copy stream — beside the
 * kernels of the batch submitted before — packs it, queues its solver launches and the copies of its results (on a second copy stream, beside whatever is
 * submitted next) and returns a ticket; avk_wait returns when the results are in `out`'s arrays and frees the ticket.  Tickets may be waited for in any order;
 * at most four are in flight.  The batch's arrays may be reused as soon as the submit has returned; `out`'s arrays belong to the library until the wait returns.
 * Every array of `batch` and `out` must be pinned (avk_host_alloc) for the copies to run beside anything: a batch with a pageable array, with sequence outputs
 * or compact BASEPAIR groups is solved inside the submit (its ticket is complete).  The submit blocks once, while the packer's plan comes back to the host —
 * behind the batch submitted before.  Results are those of avk_compare_packed, bit for bit (tests/test_gpu_async.py). */
/* ---- several GPUs: regions are independent, so a job is cut by ONE rule — shard = hash(region_id) % ranks (region ids are sequential in genome
 * order, region_generation.rs:403-409, the hash spreads dense loci over the ranks) — every rank solves its shard, and the job tally is one all-reduce
 * (SummaryWriter::add_comparison_benchmark over all regions, writers/summary.rs:146-163).  aardvark_amd/dist.py states the same hash for the Python side. */
static inline uint64_t avk_region_hash(uint64_t x) { /* splitmix64's finaliser */
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
static inline uint32_t avk_region_shard(uint64_t region_id, uint32_t ranks) { return (uint32_t)(avk_region_hash(region_id) % ranks); }
/* the regions of `whole` that rank `rank` of `world` owns (region_id[r], or first_id + r when region_id is NULL), gathered into a packed batch of their own in the
 * whole batch's order; avk_packed_shard_scatter writes a result batch of the shard into the arrays of the whole batch (per-region arrays by region, per-call arrays
 * by call; the tally is left to the caller: add the ranks' tallies, or avk_tally_allreduce) */
typedef struct avk_packed_shard avk_packed_shard;
int  avk_packed_shard_make(const avk_packed_batch *whole, const uint64_t *region_id, uint64_t first_id, uint32_t rank, uint32_t world, avk_packed_shard **out);
const avk_packed_batch *avk_packed_shard_batch(const avk_packed_shard *s);
uint64_t avk_packed_shard_regions(const avk_packed_shard *s, const uint64_t **index_in_whole);
int  avk_packed_shard_scatter(const avk_packed_shard *s, const avk_result_batch *shard_results, avk_result_batch *whole_results);
void avk_packed_shard_free(avk_packed_shard *s);
/* tally[AVK_TALLY_LEN] (host) summed over the ranks of an RCCL communicator (ncclComm_t), on the context's stream: the job's one collective.  Every rank calls it
 * with its own context and communicator; RCCL is looked up in the process (the caller that made the communicator loaded it), not linked. */
int  avk_tally_allreduce(avk_ctx *ctx, void *nccl_comm, uint64_t *tally);
/* the same collective for any block of `n` 64-bit sums (avk_tally_allreduce is this with n = AVK_TALLY_LEN; a sharded merge sums its summary counters with it) */
int  avk_counts_allreduce(avk_ctx *ctx, void *nccl_comm, uint64_t *counts, uint64_t n);

typedef struct avk_ticket avk_ticket;
int  avk_compare_packed_submit(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out, avk_ticket **ticket);
int  avk_wait(avk_ctx *ctx, avk_ticket *ticket);
int  avk_batch_upload_packed(avk_ctx *ctx, const avk_packed_batch *batch, avk_dev_batch **out);

/* The same in three steps, for callers that keep batches resident in HBM. */
int  avk_batch_upload(avk_ctx *ctx, const avk_region_batch *batch, avk_dev_batch **out);
/* tally_dev: optional device pointer (uint64[AVK_TALLY_LEN]) that receives the batch tally,
 * e.g. a tensor that is then reduced over RCCL; may be NULL. Asynchronous on the context stream. */
int  avk_compare_resident(avk_ctx *ctx, avk_dev_batch *db, const avk_compare_config *cfg, void *tally_dev);
int  avk_results_download(avk_ctx *ctx, avk_dev_batch *db, avk_result_batch *out);
void avk_batch_free(avk_ctx *ctx, avk_dev_batch *db);
int  avk_synchronize(avk_ctx *ctx);

/* bytes needed per sequence slot of region r of `batch` (upper bound of any haplotype length) */
uint32_t avk_seq_stride(const avk_region_batch *batch, uint64_t r);

/* measurement hooks: milliseconds of the solver kernels of the last avk_compare_resident /
 * avk_compare_batch on this context, from hipEvents on the context stream; how many regions
 * each workspace tier solved; algorithmic bytes of the batch (DESIGN.md "bytes per region") */
int  avk_last_kernel_ms(avk_ctx *ctx, float *ms);  /* the dominant launch: first pass of avk_region_kernel_lds */
int  avk_last_solver_ms(avk_ctx *ctx, float *ms);  /* all solver launches of the call (tier passes + tally reduce) */
int  avk_last_tier_counts(avk_ctx *ctx, uint64_t counts[5]); /* regions finished per tier, then capacity failures */
/* Pinned host memory for batch and result arrays (hipHostMalloc behind it; avk_host_free gives it back).  avk_compare_batch / avk_batch_upload /
 * avk_results_download copy arrays that live in such memory by DMA straight from / into them; arrays anywhere else go through a pinned bounce
 * buffer that the library's host threads fill or drain (a whole-genome batch: about 0.5 GB in, 0.1 GB out).  The Rust side would back its
 * FlatBatch vectors with this allocator. */
void *avk_host_alloc(avk_ctx *ctx, size_t bytes);
void  avk_host_free(avk_ctx *ctx, void *p);
/* optional: pins the bounce buffer avk_compare_batch needs for PAGEABLE arrays of a batch of up to n_regions / n_variants (kept by the context, only
 * grows), e.g. while a tool is still reading its inputs; without it the first large call pays for it */
int  avk_ctx_reserve(avk_ctx *ctx, uint64_t n_regions, uint64_t n_variants);
/* optional, for a process that makes ONE large call (the reference's `aardvark compare` is one run per process, src/main.rs:30): what the first call of a context
 * pays for beyond the call itself — the device code brought in, the table of the looked-up class, the bounce buffer of avk_ctx_reserve, the workspaces of a batch
 * of about n_regions regions — done now, e.g. on a thread beside the parsing of the inputs.  Changes no result; without it the first call does the same on demand. */
int  avk_ctx_warmup(avk_ctx *ctx, uint64_t n_regions_hint, uint64_t n_variants_hint);
int  avk_last_compare_was_one_shot(avk_ctx *ctx);  /* 1: the batch of the last avk_compare_batch / avk_optimize_pairs_batch was packed on the device
                                                      (context option device_pack) */
int  avk_last_lane_ms(avk_ctx *ctx, float *ms);    /* start of the call to the end of its lane-per-region launches (0: it had none) */
int  avk_last_lane_solved(avk_ctx *ctx, uint64_t *count); /* regions the lane-per-region kernel finished (last downloaded step);
                                                             they are not counted in any workspace tier */
int  avk_last_wide_solved(avk_ctx *ctx, uint64_t *count); /* the same for the wave-cooperative kernel of the large searches on small windows (option wide_kernel) */
/* Host utility (no GPU involved): the full GroupTypeMetrics block of region r (13 groups x 22 counters, the layout of group_metrics) from the batch, the
 * per-call outputs var_expected / var_observed of `res` and the region's compact BASEPAIR groups: the GT / HAP / WEIGHTED_HAP counters follow from
 * (expected, observed) per call (grouped_metrics.rs:183-227 and the swap of :268-277), RECORD_BP from the calls' zygosities and raw_allele_space
 * (waffle_solver.rs:455-522).  Only meaningful for regions with status 0.  Returns 0, or AVK_E_ARG. */
int avk_group_metrics_from_compact(const avk_region_batch *batch, uint64_t r, const avk_result_batch *res, uint32_t *out /* [AVK_N_GROUPS * AVK_N_FIELDS] */);
/* Host utility (no GPU involved): the wide arrays of `wide` (whichever of status, ed_h1, ed_h2, n_optima, type_present, var_expected, var_observed, var_class,
 * var_zyg are not NULL) from packed->region_packed / packed->var_packed and the batch.  Calls that no region of the batch owns are left as they are.
 * Returns 0, or AVK_E_ARG (a wide per-region array is wanted without region_packed, a per-call one without var_packed, or a region's call range is outside the
 * batch). */
int avk_results_expand(const avk_region_batch *batch, const avk_result_batch *packed, avk_result_batch *wide);
uint64_t avk_algorithmic_bytes(const avk_region_batch *batch);
/* the same with (1) or without (0) the per-region BASEPAIR groups among the outputs: a run that produces per-region records, per-call decisions and
 * the batch tally only (emit_group_metrics 0) writes no per-region groups */
uint64_t avk_algorithmic_bytes_ex(const avk_region_batch *batch, int with_groups);

/* The device aligner by itself: a batch of DWFALite SCRIPTS (reference src/dwfa/dynamic_wfa.rs:23-276).  Script s works on the two
 * byte strings bytes[base_off[s]..] ("baseline") and bytes[other_off[s]..] ("other") and makes the calls step_off[s] .. step_off[s+1]:
 * step_op 0 = update(baseline[..step_blen], other[..step_olen]) (:68-84), 1 = finalize(..) (:183-198) on ONE aligner state, as the
 * reference's tests do (dynamic_wfa.rs:283-468).  Outputs per step: edit_distance() after the call and a status (0; 2 = the aligner was
 * already finalized, :69-71; AVK_ST_CAPACITY = more than wf_cap wavefront entries; engine 1 only: AVK_ST_INVALID_INPUT for symbols other
 * than ACGT or strings over 192 bases); optionally the final wavefront (final_wf [n][wf_cap], final_wf_len [n]).
 * engine 0: one wavefront per script, the lane-group aligner of the wave-per-region kernels; engine 1: one lane per script, the 2-bit
 * aligner of the lane-per-region kernel. */
int  avk_dwfa_script_batch(avk_ctx *ctx, int engine, uint32_t n_scripts, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *base_off,
                           const uint64_t *other_off, const uint64_t *step_off, const uint8_t *step_op, const uint32_t *step_blen,
                           const uint32_t *step_olen, uint32_t *step_ed, int32_t *step_status, uint32_t wf_cap, uint32_t *final_wf,
                           uint32_t *final_wf_len);
/* profiling builds of the library only (-DAVK_PHASE_TIMING): summed clock ticks per solver phase of the last download:
 * [0] stage, [1] search A, [2] search B, [3] metrics setup, [4] base-pair metrics, [5] record metrics, [6] whole region,
 * [7] region count, [8..13] inside search A: setup, pop + quota, finalise, clone, extend, push */
int  avk_debug_phase_cycles(avk_ctx *ctx, uint64_t out[16]);

/* diagnostic (tools/gpu_hang_probe.py): busy[i] = 1 while stream i of the context still has queued work (0 the caller's stream, 1-2 the
 * solo launches' streams, 3-4 the lane launches' streams; -1 = no such stream), and the first n_counters words of the batch's device
 * counters (work-list claims, list lengths, tile claims) as they are at the time of the call, while the launches may still be running */
int  avk_debug_snapshot(avk_ctx *ctx, avk_dev_batch *db, uint32_t *counters, uint32_t n_counters, int32_t busy[5]);

/* measurement aid (bench.py's roofline by launch class): the work order of a device-packed batch — region order[k] is record k of the order the launches take their regions
 * in — and the plan's counts: counts[0..3] = regions of class C, of those not the wide kernel's, of class B, of all lane classes; counts[4 + 3 fc .. 6 + 3 fc] = first record,
 * regions and head regions of lane class fc (0 .. 5; 5 = the looked-up pairs).  Between class B and the first lane class lies the bulk.  order may be NULL. */
int  avk_debug_work_order(avk_ctx *ctx, avk_dev_batch *db, uint32_t *order, uint64_t counts[22]);

/* Stratified tallies on the device (SummaryWriter::add_comparison_benchmark with the region's containment labels,
 * src/writers/summary.rs:146-163): after avk_compare_resident with the option emit_group_metrics set, label l's block of
 * AVK_TALLY_LEN words gets the sum of the metric blocks of the solved regions whose label list names l.  The labels of region r (caller
 * order) are label_idx[label_off[r] .. label_off[r + 1]) — what avf_strat_batch_labels of the feeder library produces.  The sums are
 * ADDED to out[n_labels * AVK_TALLY_LEN] (a job sums over its batches); the per-region blocks never leave the GPU. */
int  avk_label_tallies(avk_ctx *ctx, avk_dev_batch *db, uint32_t n_labels, const uint64_t *label_off, const uint32_t *label_idx, uint64_t *out);

/* Merge path (src/merge_solver.rs:137-143): for pair p, optimize_sequences(set a, set b) and
 * report all_opt_haps[0].is_exact_match().  Pair p compares variant ranges
 * [t_off,t_cnt) vs [q_off,q_cnt) of region p of `batch` exactly like a CompareRegion. */
int  avk_optimize_pairs_batch(avk_ctx *ctx, const avk_region_batch *batch, uint32_t max_branch_factor,
                              int32_t *status, uint8_t *is_exact_match);

/* solve_merge_region (src/merge_solver.rs:110-200) for regions with k inputs each.  The expensive part is the all-pairs
 * exact-match test above; the classification on top of the pair matrix is host logic:
 *   MergeClassification (src/data_types/merge_benchmark.rs:5-14) -> AVK_MERGE_*; `members` = bit i set for the input indices
 *   of NoConflict / MajorityAgree, or the selected index itself for ConflictSelection. */
enum {
    AVK_MERGE_DIFFERENT = 0, AVK_MERGE_IDENTICAL = 1, AVK_MERGE_NO_CONFLICT = 2, AVK_MERGE_MAJORITY_AGREE = 3,
    AVK_MERGE_CONFLICT_SELECTION = 4
};
typedef struct avk_merge_config { /* MergeConfig (src/merge_solver.rs:62-84) */
    uint32_t max_branch_factor;      /* default 50 */
    uint32_t no_conflict_enabled;
    uint32_t majority_voting_enabled;
    int32_t  conflict_selection;     /* -1 = None */
} avk_merge_config;
/* A batch of MultiRegions (src/data_types/multi_region.rs): region m, input i owns variants
 * [in_off[m*k + i], +in_cnt[m*k + i]) of the variant arrays (same meaning as in avk_region_batch). */
typedef struct avk_multi_batch {
    uint64_t n_regions;
    uint32_t n_inputs;               /* k, 2..64 */
    const uint64_t *region_id;
    const uint32_t *contig_idx;
    const uint64_t *start, *end;
    const uint64_t *in_off;          /* [n_regions * k] */
    const uint32_t *in_cnt;          /* [n_regions * k] */
    uint64_t n_variants;
    const uint64_t *var_pos;
    const uint8_t  *var_type, *var_zyg;
    const uint32_t *var_raw_space;
    const uint64_t *a0_off; const uint32_t *a0_len;
    const uint64_t *a1_off; const uint32_t *a1_len;
    const uint8_t  *allele_bytes; uint64_t allele_bytes_len;
} avk_multi_batch;
/* The same batch in the PACKED form (round 3; avk_packed_batch's rules): every offset implied by order — the calls of region m follow those of region
 * m - 1, input by input; the alleles of call v follow those of call v - 1 (allele0, then allele1); call positions are relative to their region's start.
 * 8 + k bytes per region and 5 per call plus the allele bytes: a three-caller whole genome crosses PCIe as 0.13 GB instead of 0.72.  Constraints: windows
 * shorter than 65,536 bases, at most 255 calls per region and input, alleles of at most 255 bases, fewer than 2^32 calls and allele bytes, contigs shorter
 * than 4 Gbp, at most 65,535 contigs. */
typedef struct avk_packed_multi_batch {
    uint64_t n_regions;
    uint32_t n_inputs;               /* k, 2..64 */
    const uint16_t *contig_idx;      /* [n_regions] may be NULL (contig 0) */
    const uint32_t *start;           /* [n_regions] */
    const uint16_t *len;             /* [n_regions] end - start */
    const uint8_t  *in_cnt;          /* [n_regions * k] */
    uint64_t n_variants;             /* = sum of in_cnt */
    const uint16_t *var_rel_pos;     /* [n_variants] position - the region's start */
    const uint8_t  *var_type_zyg;    /* [n_variants] AVK_VT_* | AVK_ZYG_* << 4 */
    const uint8_t  *a0_len, *a1_len; /* [n_variants] */
    const uint32_t *var_raw_space;   /* [n_variants] may be NULL (= the longer allele) */
    const uint8_t  *allele_bytes; uint64_t allele_bytes_len; /* = sum of a0_len + a1_len */
} avk_packed_multi_batch;
/* Host only (no GPU): classification from the pair results.  The pairs of region m are the k(k-1)/2 pairs (i < j) in
 * lexicographic order starting at m * k(k-1)/2; has_unknown_zyg[m] != 0 = some variant of the region has an Unknown zygosity
 * (variant_delta_length bails first, :119-124 -> status AVK_ST_BAD_ZYGOSITY). */
int avk_merge_classify(uint64_t n_regions, uint32_t n_inputs, const uint32_t *in_cnt, const uint8_t *has_unknown_zyg,
                       const int32_t *pair_status, const uint8_t *pair_exact, const avk_merge_config *cfg,
                       int32_t *status, uint8_t *classification, uint64_t *members);
/* All of solve_merge_region for a batch: pairs on the GPU, classification on the host. */
int avk_merge_batch(avk_ctx *ctx, const avk_multi_batch *batch, const avk_merge_config *cfg,
                    int32_t *status, uint8_t *classification, uint64_t *members);
/* the same for a batch in the packed form (offsets by two prefix sums on the device, one kernel that writes the wide arrays there) */
int avk_merge_packed(avk_ctx *ctx, const avk_packed_multi_batch *batch, const avk_merge_config *cfg,
                     int32_t *status, uint8_t *classification, uint64_t *members);

/* ---- merge on several GPUs (BASELINE configs[4]): merge regions are mapped exactly like compare regions (src/main.rs:463-478), so the same rule cuts a packed
 * multi-region batch: avk_packed_multi_shard_make gathers the regions rank `rank` of `world` owns (avk_region_shard of region_id[r], or of first_id + r), in the
 * whole batch's order; _scatter writes the shard's status / classification / members at the regions' places in the whole batch's arrays.  The only state a merge
 * keeps across regions is MergeSummaryWriter's map (merge reason with its indices, variant type, input) -> (pass, fail) variant counts
 * (src/writers/merge_summary.rs:12-18, filled by add_merge_benchmark :57-81): avk_merge_counts ADDS a solved batch to a dense block of avk_merge_counts_len(k)
 * sums — entry ((reason * AVK_N_VARIANT_TYPES + type) * k + input) * 2 + (0 pass | 1 fail), reason = avk_merge_counts_reason() — which the ranks sum with one
 * avk_counts_allreduce (RCCL) or on the host, and avf_write_merge_summary_counts of the feeder library writes as the reference's table.  Dense blocks exist for
 * k <= AVK_MERGE_COUNTS_MAX_INPUTS inputs (the reasons carry a subset of the inputs: 2 + 2 * 2^k + k of them); avk_merge_counts_len is 0 beyond, and a job with
 * more inputs writes its summary from the scattered per-region arrays (avf_write_merge_summary). */
typedef struct avk_packed_multi_shard avk_packed_multi_shard;
int  avk_packed_multi_shard_make(const avk_packed_multi_batch *whole, const uint64_t *region_id, uint64_t first_id, uint32_t rank, uint32_t world,
                                 avk_packed_multi_shard **out);
const avk_packed_multi_batch *avk_packed_multi_shard_batch(const avk_packed_multi_shard *s);
uint64_t avk_packed_multi_shard_regions(const avk_packed_multi_shard *s, const uint64_t **index_in_whole);
int  avk_packed_multi_shard_scatter(const avk_packed_multi_shard *s, const int32_t *status, const uint8_t *classification, const uint64_t *members,
                                    int32_t *whole_status, uint8_t *whole_classification, uint64_t *whole_members);
void avk_packed_multi_shard_free(avk_packed_multi_shard *s);
#define AVK_MERGE_COUNTS_MAX_INPUTS 10
uint64_t avk_merge_counts_len(uint32_t n_inputs);
uint32_t avk_merge_counts_reason(uint32_t n_inputs, uint8_t classification, uint64_t members);
int  avk_merge_counts(const avk_packed_multi_batch *batch, const int32_t *status, const uint8_t *classification, const uint64_t *members, uint64_t *counts);

/* Host utility (no GPU involved): unit-cost edit distance of two byte strings, the value of the reference's
 * wfa_ed (src/util/sequence_alignment.rs:9-13).  The batch packer uses it for Variant::alt_ed
 * (src/data_types/variants.rs:413-415), which travels to the device with the region records. */
uint64_t avk_edit_distance(const uint8_t *a, uint64_t a_len, const uint8_t *b, uint64_t b_len);

const char *avk_version(void);
/* the build's identity: the first 16 hex digits of the SHA-256 of the kernel and host sources the library was compiled from (csrc/Makefile), "unknown" for a
 * build made another way.  Measurement files (profiles/rNN_pmc_traffic.json) carry it; bench.py reports counter traffic only when it matches the loaded library. */
const char *avk_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* AARDVARK_AMD_H */
