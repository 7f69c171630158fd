/*
 * avk_dev_types.h — layout of a region batch in HBM and of the kernel arguments.
 * Shared by the host packer (avk_pack.h), the kernels (avk_solver.inl) and the lane emulator.
 */
#ifndef AVK_DEV_TYPES_H
#define AVK_DEV_TYPES_H

#include <stdint.h>
#include "../../include/aardvark_amd.h"

/* One region (CompareRegion, reference src/data_types/compare_region.rs:13-26): one 64-byte record.
 * Everything the solver needs about the region's variants is prepared by the host packer as ONE contiguous
 * blob that the wave copies into its workspace as it is (see AvkBlobVar and pack_batch in avk_pack.h). */
struct AvkDevRegion {
    uint64_t ref_off;    /* offset of the window start in the concatenated reference bytes */
    uint32_t len;        /* window length L = end - start */
    uint32_t v_off;      /* first per-variant output word; the region owns t_cnt + q_cnt of them, truth first */
    uint32_t t_cnt;
    uint32_t q_cnt;
    uint32_t pre_status; /* low 16 bits: host validation: 0, AVK_ST_INVALID_INPUT / AVK_ST_BAD_ZYGOSITY, or AVK_PRE_SKIP_OK (pairs
                            mode: answered on the host, report status 0 / not exact); high 16 bits: bit t set = a variant of
                            type t (AVK_VT_*) is present */
    uint32_t seq_stride; /* bytes per output sequence slot (0 = no sequence output) */
    uint64_t seq_off;    /* offset of the region's 5 slots in the sequence output */
    uint32_t blob_off;   /* the region's blob starts at 8 * blob_off bytes of the blob arena */
    uint32_t blob_bytes; /* multiple of 8 */
    uint32_t alle_bytes; /* bytes of the allele section of the blob (before padding) */
    uint32_t grow;       /* max over the two sides of sum(max(0, a1_len - a0_len)): bound of a haplotype's growth */
    uint32_t orig;       /* the region's index in the caller's batch: the records are uploaded in WORK ORDER (plan_work_order),
                            so a wave gets its record straight from its work index; outputs are written at `orig` */
    uint32_t ed_bound;   /* sum of alt_ed over the region's variants: no wavefront of the region can pass this distance
                            (every haplotype is within its side's sum of the reference window), so the LDS tiers size
                            their wavefronts by min(tier cap, ed_bound) */
};

/* One variant inside a region blob (reference src/data_types/variants.rs:73-91), window-relative.
 * Blob layout (every section padded to 16 bytes):
 *   AvkBlobVar[N]   truth records then query records
 *   allele bytes    allele0 then allele1 of every variant, at a_off
 *   AvkOrdVar[N]    the variants again in the order the searches walk them — order_variants (query_optimizer.rs:372-381):
 *                   stable sort by position of [truth.., query..] — each with the position of the NEXT one (the sync
 *                   point of the step): a pop reads one 32-byte record instead of chasing order -> variant -> next variant
 *   u32 counts[8]   for the 8 types add_basepair_stats filters by (waffle_solver.rs:383-440), in the order of
 *                   AVK_SUP_TYPES: truth count | query count << 16 */
struct AvkBlobVar {
    uint32_t rel_pos;   /* position - window start */
    uint32_t a0_len;    /* ref_len() */
    uint32_t a1_len;
    uint32_t a_off;     /* allele0 bytes at a_off of the blob's allele section, allele1 at a_off + a0_len */
    uint32_t raw_space; /* raw_allele_space */
    uint32_t alt_ed;    /* Variant::alt_ed = wfa_ed(allele0, allele1) (variants.rs:413-415) */
    uint32_t type_zyg;  /* AVK_VT_* | AVK_ZYG_* << 8 */
};

struct AvkOrdVar {
    uint32_t rel_pos, a0_len, a1_len, a_off, alt_ed, type_zyg;
    uint32_t sync; /* rel_pos of the next variant in this order, or the window length for the last one */
    uint32_t vi;   /* index into AvkBlobVar[] (vi < t_cnt: a truth variant) */
};

/* host-side view of a variant (planning, scatter maps); not uploaded */
struct AvkDevVariant {
    uint32_t rel_pos;
    uint32_t a0_len;
    uint32_t a1_len;
    uint32_t a_off;     /* into PackedBatch::alleles */
    uint32_t raw_space;
    uint8_t type;
    uint8_t zyg;
    uint16_t pad;
};

#define AVK_PRE_SKIP_OK 0x1000u

/* ---- fast records: the input of the lane-per-region kernel (avk_lane.inl) ------------------------------------------------
 * Regions of the small classes (at most 3 calls per side, short window, small edit-distance bound, ACGT-only ALT alleles) get,
 * besides their AvkDevRegion, a fixed-size record of AVK_FAST_WORDS words.  Records are stored in TILES of 64: word w of the
 * record of lane l of tile t sits at [(t * AVK_FAST_WORDS + w) * 64 + l], so the 64 lanes of a wave read their records with
 * fully coalesced loads.
 *   word 0   index of the packed reference word that holds the window's first base (ref_off >> 4)
 *   word 1   ref_off & 15 | L << 4 | T << 12 | Q << 14 | order << 16   (order: bit d set = search depth d handles the next QUERY call,
 *            clear = the next truth call; order_variants query_optimizer.rs:372-381); 0xFFFFFFFF = no region in this lane
 *   word 2   first per-variant output word (AvkDevRegion::v_off)
 *   word 3   the region's index in the caller's batch (AvkDevRegion::orig)
 *   then 2 * maxv call slots of 4 words (maxv = calls per side of the record's class; truth slots first, then query slots):
 *            rel_pos | a0_len << 8 | a1_len << 16 | type << 24 | zyg << 28;  alt_ed | raw_space << 8;  allele1, 2 bits per
 *            base, 16 bases per word (2 words) */
#define AVK_FAST_HDR 4
#define AVK_FAST_WORDS 28 /* largest record: 6 call slots */
#define AVK_FAST_MAXV 3   /* most calls per side a lane takes */
/* words of a record of a class with `maxv` calls per side: header + 2 * maxv call slots (truth slots first); the tiles of a class are
 * contiguous, tile t of the class at [t * words * 64, (t + 1) * words * 64) of the class's part of the record array */
#define AVK_FAST_WORDS_OF(maxv) (AVK_FAST_HDR + 8u * (maxv))
#define AVK_FAST_CLASSES 6
#define AVK_FAST_GENERIC 5 /* classes 0 .. 4: solved by the search of avk_lane.inl; the class behind them is looked up (avk_pairs.inl) */
#define AVK_FAST_PAIR 5
/* capacities of the launch classes: sequence words (16 bases each), calls per side, wavefront cap, queue entries.
 * ed_max caps the edit distance a lane follows, not what the region may contain: with the lazily evaluated search (avk_lane.inl,
 * phaseA) the alignments of wrongly phased branches stop at their first edit, so a region of matching 40-base deletions never needs
 * a wavefront; a lane that does need more — a missed call of 7+ bases on the best path — hands its region to the wave-per-region
 * kernels, whose lanes work on the diagonals of one wavefront in parallel. */
struct AvkFastClass {
    uint32_t W, maxv, ed_max, qcap;
};
static const AvkFastClass AVK_FAST_CLASS[AVK_FAST_CLASSES] = {
    {7, 1, 6, 4},    /* one call per side, window + growth <= 112 bases (84 % of a genome) */
    {12, 1, 6, 4},   /* one call per side, <= 192 bases */
    {10, 2, 6, 16},  /* two calls per side, <= 160 bases */
    {12, 2, 6, 16},  /* two calls per side, <= 192 bases */
    {12, 3, 6, 32},  /* three calls per side, <= 192 bases (nine in ten of the regions that are left) */
    {12, 1, 0, 0},   /* AVK_FAST_PAIR: one SNV per side, the same one (seven in ten regions of a genome): no search, avk_pairs.inl */
};

#if defined(__HIPCC__) && !defined(AVK_EMU)
#define AVK_TYPES_HD __host__ __device__ static inline
#else
#define AVK_TYPES_HD static inline
#endif
/* The HEAD of a lane class (the tiles of regions with estimated edits, sorted most expensive first) is dealt out over its claims of `w`
 * records: sorted position p -> slot (p mod claims) * w + p / claims, so that every claim holds one of the `claims` most expensive
 * regions, one of the next `claims`, ... instead of the first claim holding the w most expensive ones.  Lanes that diverge take
 * turns: a claim lasts as long as the SUM of its lanes, and the launch as long as its slowest claim. */
AVK_TYPES_HD uint32_t avk_stripe_slot(uint32_t p, uint32_t head_slots, uint32_t w) {
    if (p >= head_slots || w == 0 || head_slots < w) return p;
    const uint32_t claims = head_slots / w;
    return (p % claims) * w + p / claims;
}
/* slots of a class's striped head (0: none): whole tiles of 64 that hold the n_heavy leading regions, when records are left behind them */
AVK_TYPES_HD uint32_t avk_head_slots(uint32_t maxv, uint32_t n_fast, uint32_t n_heavy, uint32_t w) {
    const uint32_t head_tiles = (n_heavy + 63u) / 64u, tiles = (n_fast + 63u) / 64u;
    return (w && w < 64u && maxv <= 2u && head_tiles > 0 && head_tiles < tiles) ? head_tiles * 64u : 0u;
}

/* Regions with this many unphased heterozygous calls (both sides counted) are big phasing searches whatever their size: every such call doubles the
 * orientations the search keeps alive (query_optimizer.rs:269-293), six of them already mean more nodes than a lane's queue ids or a 40 KB LDS slice
 * hold.  Both packers put them into class C (solved from the start of the step on a wave and an HBM slice of their own) and keep them out of the lanes'
 * three-call class — they used to be tried there or in an LDS slice first and started over late in the step, on its critical path. */
#define AVK_HET_SEARCH_MIN 6

#define AVK_WIDE_ED_MAX 62u /* largest distance a search node's wavefront holds there (one byte per offset, a state names at most 2 x 62 + 2 of them) */
/* What a region record says about whether the wave-cooperative kernel of avk_wide.inl can take the region (it looks at the alleles and the reference window
 * itself): at most 8 calls on a side, window + growth + edit bound within a byte — every offset of its wavefronts is one — and an edit bound its search nodes hold.  A launch of the wave-per-region
 * kernel with AvkKernelArgs::only_not_wide takes the records that fail this from the list the wide launch walks at the same time (it skips them). */
AVK_TYPES_HD bool avk_wide_static_ok(uint32_t len, uint32_t grow, uint32_t ed_bound, uint32_t t_cnt, uint32_t q_cnt, uint32_t pre_status) {
    return !(pre_status & 0xFFFFu) && t_cnt + q_cnt != 0 && t_cnt <= 8u && q_cnt <= 8u && (uint64_t)len + grow + ed_bound <= 255ull && ed_bound <= AVK_WIDE_ED_MAX;
}

#define AVK_CAP_BOUND_ONLY 0x80000000u /* flag in AvkTier::ed_cap: the cap is only taken by regions whose own bound is below it (avk_solver.inl) */
/* capacities of one workspace tier */
struct AvkTier {
    uint64_t ws_bytes; /* bytes of workspace per wave in this tier */
    uint32_t ed_cap;   /* 0 = exact worst case (wavefront can hold any edit distance) */
    uint32_t pad;
};

/* partial-tally geometry: AVK_TALLY_LEN sums + 5 tier counters + 8 profiling words, padded */
#define AVK_TALLY_STRIDE 320
#define AVK_TALLY_LANE_SOLVED 310 /* word of a partial tally: regions finished (either way) by the lane-per-region kernel */
#define AVK_TALLY_WIDE_SOLVED 311 /* regions finished (either way) by the wave-cooperative kernel of avk_wide.inl */
#define AVK_TALLY_COPIES 64
/* tail of a bulk workgroup's LDS: 16 control words + AVK_TALLY_LEN (rounded up) tally words */
#define AVK_WG_TAIL_BYTES (64 + 4 * 288)

struct AvkKernelArgs {
    /* inputs */
    const AvkDevRegion *regions;
    const uint32_t *blob;      /* region blobs, see AvkBlobVar */
    const uint8_t *ref_bytes;  /* concatenated contigs, one byte per base (used for windows that hold non-ACGT symbols) */
    const uint32_t *ref_2bit;  /* the same, 16 bases per word, base i of a word in bits 2i..2i+1 (A 0, C 1, G 2, T 3); may be NULL */
    const uint32_t *ref_exc;   /* bit w of this bitmap: packed word w holds a symbol other than upper-case A/C/G/T */
    uint32_t n_regions;
    uint32_t max_branch_factor;
    uint32_t enable_exact_shortcut;
    uint32_t mode; /* 0 solve_compare_region; 1 merge pairs: optimize_sequences only, report all_opt_haps[0].is_exact_match() */
    uint32_t pad0_;
    uint32_t pass_tier;  /* workspace tier of this launch: 0 small LDS slice, 1 large LDS slice, 2 per-wave HBM slice, 3 big HBM slice */
    /* work distribution */
    const uint32_t *work_list; /* NULL = records work_base .. work_base + n_work - 1; else record indices (overflow pass) */
    uint32_t work_base;
    uint32_t pad2_;
    const uint32_t *n_work_dev; /* when set, the number of work items is read from device memory (overflow pass) */
    uint32_t *work_counter;    /* 8 claim counters, 32 words (128 B) apart, one per shard of the dynamic part of the work list */
    uint32_t n_waves;          /* persistent waves of this launch (solo waves not counted) */
    uint32_t n_work;           /* length of the work list when n_work_dev is NULL */
    uint32_t static_pct;       /* share of the work list dealt statically (item k to wave k mod n_waves), the rest is claimed */
    uint32_t n_shards;         /* claim counters in use (1..8) */
    uint32_t claim;            /* regions per claim */
    uint32_t esc_bytes;        /* bulk launch (workgroups of exactly 4 waves): offset of the workgroup's TAIL in its LDS = 4 slices of
                                  tier[0].ws_bytes.  The tail holds 16 control words and the workgroup's tally (AVK_WG_TAIL_BYTES):
                                  - the waves add a region's nonzero counters to the LDS tally; the last wave to leave flushes it to
                                    a partial tally in HBM (a global atomic per counter and region was 16 MB of HBM writes per launch);
                                  - with esc_enabled, a wave whose region outgrows its slice takes the workgroup's whole LDS
                                    ([0, esc_bytes)) while its sibling waves park between regions, and solves it again at once with
                                    tier 1's cap — no prediction, no later launch.  0 = no tail (solo and HBM launches). */
    uint32_t esc_enabled;
    uint32_t high_priority;    /* raise the wave priority (the solo launch of the predicted-hard regions) */
    uint32_t *overflow_list;   /* regions that exhausted this pass's tiers */
    uint32_t *overflow_count;
    /* A second work source, taken one record at a time once the launch's own list is exhausted: records extra_base ..
     * extra_base + extra_n - 1, ticket counter extra_counter.  The HBM launch of the main stream shares the class C list
     * (and its counter) with the HBM solo launch this way: whatever the solo launch has not started when the bulk is done
     * is spread over the whole chip. */
    uint32_t *extra_counter;
    uint32_t extra_base;
    uint32_t extra_n;
    /* workspaces */
    uint8_t *hbm_ws;           /* n_waves slices of tier[2] (or tier[3]) bytes */
    /* in-place escalation of the HBM launch: a region that outgrows its wave's tier-2 slice is solved again at once in one of
     * big_slots shared tier-3 slices (claimed through big_busy[slot], one word each), so the last tier needs no launch of its own */
    uint8_t *big_ws;
    uint32_t *big_busy;
    uint32_t big_slots;
    uint32_t pad1_;
    AvkTier tier[4];
    /* outputs */
    uint32_t *region_out;    /* [n][4]: status, ed_h1, ed_h2, n_optima | type_present << 16 — one 16-byte store per region */
    uint32_t *group_metrics; /* optional [n][13][22] */
    uint32_t *var_out;       /* [n_variants]: expected | observed << 8 | class << 16 | resolved zygosity << 24 */
    uint8_t *seq_bytes;      /* optional */
    uint32_t *seq_len;
    uint64_t *tally;         /* [AVK_TALLY_COPIES][AVK_TALLY_STRIDE] partial tallies; wave w adds into copy (w / 4) % AVK_TALLY_COPIES,
                                a reduce kernel sums the copies: 4096 waves adding into ONE block serialise on its few cache lines */
    /* Device-packed batches (avk_devpack.inl) write the AvkDevRegion + blob of a lane-class region only when a wave-per-region launch takes
     * the region (0.15 % of a genome: what the lanes hand back): records from index lazy_from on are written by the wave that is about to
     * solve them, with the packer's arguments at lazy_dp (device memory).  NULL = every record exists. */
    const void *lazy_dp;
    uint32_t lazy_from;
    uint32_t pad3_;
    /* compact per-region BASEPAIR groups (optional): region r (caller order) owns groups [bp_off[r], bp_off[r + 1]) of bp_out — the joint group, then one per
     * call type present among its calls, in type order — 4 counters each (AVK_F_BP_TRUTH_TP .. AVK_F_BP_QUERY_FP).  The other 18 counters of a group follow
     * from the per-call decisions (avk_group_metrics_from_compact). */
    const uint32_t *bp_off;
    uint32_t *bp_out;
    uint32_t only_not_wide; /* this launch takes only the records avk_wide_static_ok() turns down (a launch of avk_wide.inl has the others) */
    uint32_t team;          /* 1: avk_region_kernel_team — a workgroup is ONE region's team: wave 0 runs the search, its siblings take the independent pieces it posts
                               (the two haplotypes of up to two children of a popped node, the alignments of the metrics): the launch of the long windows */
};

#endif
