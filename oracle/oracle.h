/*
 * oracle.h — C API of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain, sequential C++ restatement of the
 * reference algorithm (PacificBiosciences/aardvark v0.10.5) for the compare hot path.
 * It exists to check the HIP kernels and to serve as the timed CPU baseline of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (libaardvark_amd.so) never links, loads or calls it.
 *
 * Parity pinning: checked against every known-answer test the reference holds for this
 * path (tests/golden/ JSON files, transcribed from the #[test] functions cited there).  The
 * reference itself (Rust) cannot be built in this image, so there is no oracle/_ref.
 */
#ifndef AVK_ORACLE_H
#define AVK_ORACLE_H

#include <stdint.h>
#include "../include/aardvark_amd.h" /* POD batch/result structs and enums only */

#ifdef __cplusplus
extern "C" {
#endif

/* util/sequence_alignment.rs:9-13 and :20-51 */
uint64_t orc_wfa_ed(const uint8_t *a, uint64_t alen, const uint8_t *b, uint64_t blen);
uint64_t orc_edit_distance(const uint8_t *a, uint64_t alen, const uint8_t *b, uint64_t blen);

/* incremental DWFALite (dwfa/dynamic_wfa.rs:23-276); max_ed == UINT64_MAX means unbounded.
 * update/finalize return 0 ok, 1 MaxEditDistance, 2 AlreadyFinalized. */
void    *orc_dwfa_new(uint64_t max_ed);
void     orc_dwfa_free(void *h);
void    *orc_dwfa_clone(const void *h);
int      orc_dwfa_update(void *h, const uint8_t *base, uint64_t blen, const uint8_t *other, uint64_t olen);
int      orc_dwfa_finalize(void *h, const uint8_t *base, uint64_t blen, const uint8_t *other, uint64_t olen);
uint64_t orc_dwfa_ed(const void *h);
uint64_t orc_dwfa_wavefront(const void *h, uint64_t *out, uint64_t cap);
int      orc_dwfa_equal(const void *a, const void *b);

/* A scripted HaplotypeDWFA / ComparisonNode / ExactMatchNode session for the node-level
 * golden tests (haplotype_dwfa.rs:252-332, query_optimizer.rs:504-530,
 * exact_gt_optimizer.rs:495-523).  n_haps = 1 or 2. */
void    *orc_hapnode_new(int n_haps, uint64_t region_start, uint64_t max_ed);
void     orc_hapnode_free(void *h);
/* allele codes: 1 REF, 2 ALT (phase_enums.rs:20-24). sync < 0 means None. returns 0 ok else error;
 * *success_out receives the AND of the per-haplotype success flags */
int      orc_hapnode_extend(void *h, const uint8_t *ref, uint64_t ref_len, int is_truth, uint64_t pos,
                            const uint8_t *a0, uint64_t a0_len, const uint8_t *a1, uint64_t a1_len,
                            int allele_h1, int allele_h2, int64_t sync, int *success_out);
int      orc_hapnode_finalize(void *h, const uint8_t *ref, uint64_t ref_len, uint64_t region_end);
uint64_t orc_hapnode_ed(const void *h, int hap);
uint64_t orc_hapnode_skip(const void *h, int hap);       /* truth + query skip distance */
uint64_t orc_hapnode_cost(const void *h);                /* sum of total_cost over haps */
uint64_t orc_hapnode_seq(const void *h, int hap, int is_truth, uint8_t *out, uint64_t cap);
uint64_t orc_hapnode_alleles(const void *h, int hap, int is_truth, uint8_t *out, uint64_t cap);

/* optimize_sequences (query_optimizer.rs:166-365) on region `r` of `batch` with `ref` = the
 * full contig.  Returns the number of tied optima (or -status).  For optimum k < cap:
 * ed[2k..], skips[4k..] = truth_vs1, truth_vs2, query_vs1, query_vs2; zygosities are written
 * per variant as [k][T] / [k][Q]; sequences via orc_last_sequence(). */
int64_t  orc_optimize_sequences(const avk_region_batch *batch, uint64_t r, const uint8_t *ref, uint64_t ref_len,
                                uint32_t max_branch_factor, uint32_t cap,
                                uint64_t *ed, uint64_t *skips, uint8_t *truth_zyg, uint8_t *query_zyg);
/* sequence s (0 truth1, 1 truth2, 2 query1, 3 query2) of optimum k of the last
 * orc_optimize_sequences call on this thread */
uint64_t orc_last_sequence(uint32_t k, int s, uint8_t *out, uint64_t cap);

/* optimize_gt_alleles (exact_gt_optimizer.rs:108-357): alleles in / out are codes 1 REF, 2 ALT
 * per truth / query variant of region r.  returns num_errors or -status. */
int64_t  orc_optimize_gt_alleles(const avk_region_batch *batch, uint64_t r, const uint8_t *ref, uint64_t ref_len,
                                 const uint8_t *truth_alleles, const uint8_t *query_alleles,
                                 uint8_t *truth_out, uint8_t *query_out);

/* solve_compare_region for every region of the batch (waffle_solver.rs:122-284), `threads`
 * worker threads pulling regions dynamically (the rayon loop of main.rs:251-268).
 * `refs[c]`/`ref_lens[c]` = contig c. */
int      orc_compare_batch(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens,
                           uint32_t n_contigs, const avk_compare_config *cfg, avk_result_batch *out, int threads);

/* timed CPU baseline (bench.py cpu_baseline): reps passes over the batch on `threads` workers */
int      orc_bench(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
                   const avk_compare_config *cfg, int threads, int reps, double *seconds, uint64_t *checksum);

/* perform_basepair_compare (waffle_solver.rs:611-658): out[4] = truth_tp, truth_fn, query_tp, query_fp */
void     orc_basepair_compare(const uint8_t *ref, uint64_t rl, const uint8_t *t, uint64_t tl,
                              const uint8_t *q, uint64_t ql, uint64_t out[4]);

/* windows onto the metric containers (grouped_metrics.rs:183-277, variant_metrics.rs:43-101) and
 * generate_haplotype_sequence (waffle_solver.rs:685-778) for the golden tests */
int      orc_group_add_truth(uint64_t g[AVK_N_FIELDS], uint64_t weight, uint8_t expected, uint8_t observed);
int      orc_group_add_query(uint64_t g[AVK_N_FIELDS], uint64_t weight, uint8_t expected, uint8_t observed);
void     orc_group_swap(uint64_t g[AVK_N_FIELDS], const uint64_t other[AVK_N_FIELDS]);
int      orc_variant_metrics(uint8_t expected, uint8_t observed, uint8_t out[3], uint8_t toggled[3]);
int64_t  orc_generate_haplotype_sequence(const avk_region_batch *batch, uint64_t r, int side, const uint8_t *ref, uint64_t ref_len,
                                         const uint8_t *zygosities, int hap, uint8_t *out, uint64_t cap, uint64_t *failed_ed);

/* solve_merge_region's pairwise test (merge_solver.rs:135-147) */
int      orc_optimize_pairs_batch(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens,
                                  uint32_t n_contigs, uint32_t max_branch_factor, int32_t *status,
                                  uint8_t *is_exact_match, int threads);

/* per-region search statistics: pops A, max queue A, pops B, max queue B, tied optima, max edit distance */
int      orc_region_stats(const avk_region_batch *batch, uint64_t r, const uint8_t *ref, uint64_t ref_len, const avk_compare_config *cfg, uint64_t out[6]);

/* search statistics of the last orc_compare_batch on this process (sizing aid for the kernels):
 * [0] max pops in optimize_sequences, [1] max live queue there, [2] max pops in optimize_gt_alleles,
 * [3] max live queue there, [4] max edit distance seen by any DWFA, [5] max tied optima,
 * [6] total pops A, [7] total pops B, [8] total wfa_ed calls, [9] haplotypes whose incremental DWFA
 * distance differed from a fresh wfa_ed of the final sequences (must stay 0) */
void     orc_last_stats(uint64_t out[16]);

#ifdef __cplusplus
}
#endif
#endif
