/*
 * aardvark_feeder.h — C-ABI of libaardvark_feeder.so: the host-side feeder and summary writer around the
 * compare hot path.  Plain host C++ behind it, no GPU involved.
 *
 * What it replaces in the reference (PacificBiosciences/aardvark v0.10.5):
 *   avf_genome_load     ReferenceGenome::from_fasta                           (src/main.rs:94)
 *   avf_feed_compare    RegionIterator::new_compare_iterator + the iterator   (src/parsing/region_generation.rs:61-122, :281-478)
 *                       with load_variants_in_region / parse_variant /
 *                       parse_genotype / get_variant_type                     (:489-758)
 *                       and LoadedBed::preload_bed_file                       (src/parsing/noodles_helper.rs:48-86)
 *   avf_write_summary   SummaryWriter::write_summary                          (src/writers/summary.rs:163-395)
 *   avf_feed_merge      RegionIterator::new_merge_iterator + the iterator     (src/parsing/region_generation.rs:129-192, :281-478)
 *   avf_write_merge_*   VariantMerger, MergeSummaryWriter                     (src/writers/variant_merger.rs, merge_summary.rs)
 *
 * The feed hands out an avk_region_batch (include/aardvark_amd.h) whose regions carry the reference's region ids
 * and windows, ready for avk_compare_batch.
 */
#ifndef AARDVARK_FEEDER_H
#define AARDVARK_FEEDER_H

#include <stdint.h>
#include "aardvark_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avf_genome avf_genome;
typedef struct avf_feed avf_feed;

/* error text of the last failed call on this thread */
const char *avf_last_error(void);

/* FASTA (plain or gzip/BGZF): contig name = header up to the first white space; the sequence bytes with a-z folded to A-Z (the default of this
 * build, see avf_genome_load_case below — NOT the file's raw bytes: that is avf_genome_load_case(path, 0, out)). */
int avf_genome_load(const char *fasta_path, avf_genome **out); /* = avf_genome_load_case(path, 1, out) */
/* Case of the reference bases.  ReferenceGenome::from_fasta lives in a crate that is not vendored with the reference
 * (rust-lib-reference-genome 0.2.1; its dependencies are bio, flate2, log, rustc-hash, simple-error — Cargo.lock:1344-1353), so whether it
 * folds case cannot be read there; every test of the reference uses upper-case contigs.  GRCh38 is about half soft-masked (lower case)
 * while VCF alleles are upper case; compared as raw bytes, wfa_ed(reference window, haplotype) (waffle_solver.rs:639-656) would count a
 * case difference for every ALT base that re-states a lower-case reference base.  upper_case = 1 (the default of this build and of the
 * tools, --reference-case upper) maps a-z to A-Z while loading: IUPAC codes and N stay as they are; 0 (--reference-case raw) keeps the
 * file's bytes. */
int avf_genome_load_case(const char *fasta_path, int upper_case, avf_genome **out);
uint32_t avf_genome_n_contigs(const avf_genome *g);
const char *avf_genome_name(const avf_genome *g, uint32_t i);
const uint8_t *avf_genome_seq(const avf_genome *g, uint32_t i);
uint64_t avf_genome_len(const avf_genome *g, uint32_t i);
void avf_genome_free(avf_genome *g);

/* Name of sample `index` of a VCF's #CHROM line (get_vcf_sample_name, src/parsing/noodles_helper.rs:103-120), copied into out (cap
 * bytes, always terminated). */
int avf_vcf_sample_name(const char *vcf, uint32_t index, char *out, uint64_t cap);

/* Region generation for `compare`.  truth_sample / query_sample: NULL or "" = the first sample of the file
 * (src/cli/compare.rs:186-194).  regions_bed is required (region_generation.rs:93-97).  min_variant_gap > 0
 * (default of the reference: 50); enable_trimming = !--disable-variant-trimming.
 * The batch's contig_idx refers to the contigs of `g` in file order. */
int avf_feed_compare(const char *truth_vcf, const char *truth_sample, const char *query_vcf, const char *query_sample,
                     const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap, int enable_trimming, avf_feed **out);
const avk_region_batch *avf_feed_batch(const avf_feed *f);
/* provenance of batch variant v, for the VCF writers: 0-based index of its record among the data lines of its
 * file, and the 1-based ALT index it came from */
const uint64_t *avf_feed_var_record(const avf_feed *f);
const uint32_t *avf_feed_var_alt_index(const avf_feed *f);
/* The compare feed in the library's PACKED form (avk_packed_batch, aardvark_amd.h: 10 bytes per region, 5 per call, the allele bytes) — what
 * avk_compare_packed / avk_batch_upload_packed take.  The feed's calls already lie in the order the form implies (region by region, truth calls then query
 * calls, allele0 then allele1).  The arrays come from alloc(user, bytes) — pass avk_host_alloc and the context for pinned memory that is copied by DMA — and
 * belong to the caller; var_raw_space stays NULL when every call's raw space is its longer allele.  Returns 0; 1 when the feed does not meet the form's
 * constraints (a window of 65,536 bases or more, more than 255 calls on a side, an allele longer than 255 bases, ...: nothing is allocated, use
 * avf_feed_batch); a negative AVK_E_* code on errors. */
int avf_feed_pack(const avf_feed *f, void *(*alloc)(void *user, size_t bytes), void *user, avk_packed_batch *out);
/* Regions [first, first + n) of a packed feed as a batch of their own (its arrays point into `all`'s); *v_first = index of the part's first call in the
 * feed's call arrays: the part's per-call results are indexed from there. */
int avf_packed_slice(const avf_feed *f, const avk_packed_batch *all, uint64_t first, uint64_t n, avk_packed_batch *part, uint64_t *v_first);
/* Region generation for `merge` (RegionIterator::new_merge_iterator, region_generation.rs:129-192): the same walk over
 * n_inputs VCFs (1..64), in priority order.  samples: NULL, or per input NULL / "" = that file's first sample
 * (src/cli/merge.rs:196-198).  The feed hands out an avk_multi_batch for avk_merge_batch; avf_feed_batch is NULL for it
 * (and avf_feed_multi_batch is NULL for a compare feed). */
int avf_feed_merge(uint32_t n_inputs, const char *const *vcfs, const char *const *samples, const char *regions_bed, const avf_genome *g,
                   uint64_t min_variant_gap, int enable_trimming, avf_feed **out);
const avk_multi_batch *avf_feed_multi_batch(const avf_feed *f);
/* The merge feed in the packed form (avk_packed_multi_batch) for avk_merge_packed, and a range of its regions as a batch of its own: as avf_feed_pack /
 * avf_packed_slice for compare feeds (returns 1 when the feed does not meet the form's constraints: use avf_feed_multi_batch). */
int avf_feed_pack_multi(const avf_feed *f, void *(*alloc)(void *user, size_t bytes), void *user, avk_packed_multi_batch *out);
int avf_packed_multi_slice(const avf_feed *f, const avk_packed_multi_batch *all, uint64_t first, uint64_t n, avk_packed_multi_batch *part);
/* The two halves of a feed on their own, so that a caller can read the VCFs while the reference genome is still loading (the reference
 * preloads its variants the same way, RegionIterator::preload_all_variants, region_generation.rs:199-279):
 * avf_calls_load parses one VCF (every chromosome), avf_feed_from_calls walks the regions over already loaded call sets.
 * merge = 0: a compare feed (exactly two inputs: truth, query), merge != 0: a merge feed. */
typedef struct avf_calls avf_calls;
int avf_calls_load(const char *vcf, const char *sample, int enable_trimming, avf_calls **out);
void avf_calls_free(avf_calls *c);
uint64_t avf_calls_count(const avf_calls *c); /* loaded calls (every chromosome) */
int avf_feed_from_calls(uint32_t n_inputs, const avf_calls *const *calls, const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap, int merge,
                        avf_feed **out);
/* variants loaded per input after parsing and the chromosome-span filter (the "Loaded N truth variants" log lines) */
uint64_t avf_feed_loaded_variants(const avf_feed *f, int input);
void avf_feed_free(avf_feed *f);

/* metrics_mask: bit i set = write MetricsType i of {GT, HAP, WEIGHTED_HAP, BASEPAIR, RECORD_BP} (the reference always
 * writes GT and BASEPAIR, the others on request, in the order GT, BASEPAIR, HAP, WEIGHTED_HAP, RECORD_BP: src/main.rs:134-147).
 * tally = the ALL block of avk_result_batch::tally summed over all batches.  A path ending in .csv is comma separated. */
#define AVF_METRIC_GT 1u
#define AVF_METRIC_HAP 2u
#define AVF_METRIC_WEIGHTED_HAP 4u
#define AVF_METRIC_BASEPAIR 8u
#define AVF_METRIC_RECORD_BP 16u
int avf_write_summary(const char *path, const char *compare_label, const uint64_t *tally, uint32_t metrics_mask);

/* ---- stratifications (src/parsing/stratifications.rs) ------------------------------------------------------------------
 * avf_strat_load = Stratifications::from_tsv_batch (:30-86): a TSV of `label <TAB> BED path` rows (paths relative to the
 * TSV's folder), duplicate labels rejected; labels end up in sorted order (the reference collects them in a BTreeMap).
 * Intervals are kept 0-based inclusive per label and chromosome (:147-176). */
typedef struct avf_strat avf_strat;
int avf_strat_load(const char *tsv_path, avf_strat **out);
uint32_t avf_strat_n_labels(const avf_strat *s);
const char *avf_strat_label(const avf_strat *s, uint32_t label);
uint64_t avf_strat_n_intervals(const avf_strat *s, uint32_t label, const char *chrom); /* COITree::len of that chromosome */
/* Stratifications::containments / overlaps (:88-113) of the 0-based inclusive range [first, last] on chrom: the indices of
 * the labels with an interval that contains / overlaps it, ascending, written to out (at most cap); returns how many */
uint32_t avf_strat_containments(const avf_strat *s, const char *chrom, int64_t first, int64_t last, uint32_t *out, uint32_t cap);
uint32_t avf_strat_overlaps(const avf_strat *s, const char *chrom, int64_t first, int64_t last, uint32_t *out, uint32_t cap);
/* The labels containing region r of `batch` the way solve_compare_region asks (src/waffle_solver.rs:151-166):
 * [min first position, max over the two LAST variants of pos + ref_len) of CompareRegion::var_coordinates
 * (compare_region.rs:54-66), queried as first..last-1. */
uint32_t avf_strat_region_labels(const avf_strat *s, const avf_genome *g, const avk_region_batch *batch, uint64_t r, uint32_t *out, uint32_t cap);
/* avf_strat_region_labels for regions first .. first + n - 1 at once (several threads), as a compressed list: the labels of region
 * first + k are label_idx[label_off[k] .. label_off[k + 1]).  Two calls: with label_idx = NULL label_off[0..n] is filled
 * (label_off[n] = entries needed), with room for that many entries the list itself.  This is what avk_label_tallies takes. */
int avf_strat_batch_labels(const avf_strat *s, const avf_genome *g, const avk_region_batch *batch, uint64_t first, uint64_t n, uint64_t *label_off,
                           uint32_t *label_idx);
void avf_strat_free(avf_strat *s);
/* avf_write_summary with the stratified blocks after the ALL block (summary.rs:203-222): strat_tallies holds
 * avf_strat_n_labels(s) blocks of AVK_TALLY_LEN words, block l = the sum over the regions label l contains */
int avf_write_summary_stratified(const char *path, const char *compare_label, const uint64_t *tally, const avf_strat *s, const uint64_t *strat_tallies,
                                 uint32_t metrics_mask);

/* One of the two annotated VCFs of `compare` (VariantCategorizer, src/writers/variant_categorizer.rs:41-230; source 0 =
 * truth.vcf.gz, 1 = query.vcf.gz): the meta lines of input_vcf, the aardvark_version / aardvark_command lines and the
 * BD / EA / OA / RI FORMAT definitions, one sample column (sample_name, or the input's first sample when empty), then one
 * record per variant of every solved region (status 0) in region order:
 *   CHROM POS . REF ALT . . . GT:BD:EA:OA:RI  gt:TP|FN|FP:expected:observed:region_id
 * with REF/ALT as the solver saw them (after trimming).  Written as BGZF; a tabix index goes to out_path + ".tbi"
 * (src/writers/noodles_idx.rs:7-20).  status / var_* are the arrays of the avk_result_batch of `batch`. */
int avf_write_annotated_vcf(const char *out_path, const char *input_vcf, const char *sample_name, const char *version, const char *command_line,
                            const avf_genome *g, const avk_region_batch *batch, int source, const int32_t *status, const uint8_t *var_expected,
                            const uint8_t *var_observed, const uint8_t *var_class);

/* ---- outputs of `merge` ---------------------------------------------------------------------------------------------------
 * avf_write_merge_outputs = VariantMerger (src/writers/variant_merger.rs:47-349) over a whole job: creates out_folder and in it
 *   passing.vcf.gz          header of primary_vcf (input 0) + aardvark_version / aardvark_command + INFO SOURCES, MR + FORMAT RI,
 *                           one sample column (sample_name, or the first sample of primary_vcf when empty); for every solved
 *                           region that is not `different`, the variants of its source input (lowest member index of
 *                           no_conflict / majority, the selected index, input 0 for identical; :167-173) as
 *                             CHROM POS . REF ALT . . SOURCES=<tags of the members>;MR=<reason>  GT:RI  gt:region_id
 *   regions.bed.gz          CHROM start end <reason>_<region_id> of those regions
 *   failed_regions.bed.gz   the same for the `different` regions
 * each with a tabix index next to it (index_merger, :316-349).  Regions whose status is not 0 are written nowhere (main.rs:517-519).
 * tags[i] = the annotation tag of input i (--vcf-tag, default "vcf_<i>"); status / classification / members are the outputs of
 * avk_merge_batch for `batch`. */
int avf_write_merge_outputs(const char *out_folder, const char *primary_vcf, const char *sample_name, const char *version, const char *command_line,
                            const avf_genome *g, const avk_multi_batch *batch, const char *const *tags, const int32_t *status,
                            const uint8_t *classification, const uint64_t *members);
/* MergeSummaryWriter (src/writers/merge_summary.rs): pass / fail variant counts per (merge reason with its indices, variant type,
 * input), rows in the reference's key order (BTreeMap over (MergeClassification, VariantType, vcf_index)); columns merge_reason,
 * variant_type, vcf_index, vcf_label, pass_variants, fail_variants.  A path ending in .csv is comma separated. */
int avf_write_merge_summary(const char *path, const avk_multi_batch *batch, const char *const *tags, const int32_t *status,
                            const uint8_t *classification, const uint64_t *members);
/* the same table from the dense block of sums a merge sharded over several GPUs adds up with one all-reduce (avk_merge_counts / avk_counts_allreduce of
 * aardvark_amd.h; counts_len = avk_merge_counts_len(n_inputs)): byte for byte the table avf_write_merge_summary writes for the whole job */
int avf_write_merge_summary_counts(const char *path, uint32_t n_inputs, const char *const *tags, const uint64_t *counts, uint64_t counts_len);

/* ---- the debug tables of --output-debug, both BGZF-compressed tab-separated text, rows appended batch by batch ----------------
 * region_summary.tsv.gz   (RegionSummaryWriter, src/writers/region_summary.rs): one row per metric kind of metrics_mask and solved
 *                         region with the region's JOINT metrics; columns region_id, coordinates ("chrom:start+1-end"),
 *                         comparison, the six counts, recall / precision / F1, truth_fn_gt, query_fp_gt
 * region_sequences.tsv.gz (RegionSequenceWriter, src/writers/region_sequence.rs): region_id, coordinates, reference window and
 *                         the four haplotype sequences (needs avk_compare_config::enable_sequences)
 * first / n select the regions of `batch` the arrays belong to: status[k], group_metrics block k, sequence slots k describe
 * region first + k. */
typedef struct avf_table avf_table;
int avf_region_summary_open(const char *path, uint32_t metrics_mask, avf_table **out);
int avf_region_summary_rows(avf_table *t, const avf_genome *g, const avk_region_batch *batch, uint64_t first, uint64_t n, const int32_t *status,
                            const uint32_t *group_metrics);
int avf_region_sequences_open(const char *path, avf_table **out);
int avf_region_sequences_rows(avf_table *t, const avf_genome *g, const avk_region_batch *batch, uint64_t first, uint64_t n, const int32_t *status,
                              const uint8_t *seq_bytes, const uint32_t *seq_len, const uint64_t *seq_off, const uint32_t *seq_stride);
int avf_table_close(avf_table *t);

#ifdef __cplusplus
}
#endif
#endif /* AARDVARK_FEEDER_H */
