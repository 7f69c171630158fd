/*
 * avf_debugout.cpp — the two debug tables of `aardvark compare --output-debug`:
 *   region_summary.tsv.gz    RegionSummaryWriter  (src/writers/region_summary.rs:11-139): the joint metrics of every solved region
 *   region_sequences.tsv.gz  RegionSequenceWriter (src/writers/region_sequence.rs:9-77): reference + the four haplotype sequences
 * Both are tab-separated tables written through BGZF; rows can be appended batch by batch.  Host code only.
 */
#include "../../../include/aardvark_feeder.h"

#include "avf_bgzf.h"

#include <string>

int avf_fail_(int code, const char *fmt, ...);
std::string avf_fmt_f64_(double v); /* avf_feeder.cpp: ryu's float text */

struct avf_table {
    FILE *fp = nullptr;
    avf_bgzf::Stream *out = nullptr;
    int kind = 0; /* 0 region summary, 1 region sequences */
    uint32_t metrics_mask = 0;
    bool ok = true;
};

extern "C" {

int avf_region_summary_open(const char *path, uint32_t metrics_mask, avf_table **out) {
    if (!path || !out) return avf_fail_(AVK_E_ARG, "null argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return avf_fail_(AVK_E_ARG, "cannot create %s", path);
    avf_table *t = new avf_table();
    t->fp = fp;
    t->out = new avf_bgzf::Stream(fp);
    t->kind = 0;
    t->metrics_mask = metrics_mask;
    t->ok = t->out->write("region_id\tcoordinates\tcomparison\ttruth_total\ttruth_tp\ttruth_fn\tquery_total\tquery_tp\tquery_fp\tmetric_recall\tmetric_precision\t"
                          "metric_f1\ttruth_fn_gt\tquery_fp_gt\n");
    *out = t;
    return 0;
}

int avf_region_sequences_open(const char *path, avf_table **out) {
    if (!path || !out) return avf_fail_(AVK_E_ARG, "null argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return avf_fail_(AVK_E_ARG, "cannot create %s", path);
    avf_table *t = new avf_table();
    t->fp = fp;
    t->out = new avf_bgzf::Stream(fp);
    t->kind = 1;
    t->ok = t->out->write("region_id\tcoordinates\tref_seq\ttruth_seq1\ttruth_seq2\tquery_seq1\tquery_seq2\n");
    *out = t;
    return 0;
}

/* rows of regions [first, first + n) of `batch`; status / group_metrics (n blocks of 13 x 22) are indexed from 0 = region `first` */
int avf_region_summary_rows(avf_table *t, const avf_genome *g, const avk_region_batch *b, uint64_t first, uint64_t n, const int32_t *status,
                            const uint32_t *group_metrics) {
    if (!t || t->kind != 0 || !g || !b || !status || !group_metrics || first + n > b->n_regions) return avf_fail_(AVK_E_ARG, "null or invalid argument");
    struct Kind {
        uint32_t bit;
        const char *name;
        int base;
        bool gt;
    };
    /* the order main.rs pushes them (:134-147) */
    static const Kind kinds[5] = {{AVF_METRIC_GT, "GT", AVK_F_GT_TRUTH_TP, true},      {AVF_METRIC_BASEPAIR, "BASEPAIR", AVK_F_BP_TRUTH_TP, false},
                                  {AVF_METRIC_HAP, "HAP", AVK_F_HAP_TRUTH_TP, false},  {AVF_METRIC_WEIGHTED_HAP, "WEIGHTED_HAP", AVK_F_WHAP_TRUTH_TP, false},
                                  {AVF_METRIC_RECORD_BP, "RECORD_BP", AVK_F_RBP_TRUTH_TP, false}};
    std::string row;
    for (uint64_t k = 0; k < n && t->ok; ++k) {
        if (status[k] != 0) continue; /* failed regions are not written (compare_parallel.rs:229-262) */
        const uint64_t r = first + k;
        const uint32_t *joint = group_metrics + (size_t)k * AVK_N_GROUPS * AVK_N_FIELDS; /* group 0 */
        const std::string coords = std::string(avf_genome_name(g, b->contig_idx ? b->contig_idx[r] : 0)) + ":" + std::to_string(b->start[r] + 1) + "-" +
                                   std::to_string(b->end[r]); /* Coordinates' Display, coordinates.rs:73-78 */
        for (const Kind &kd : kinds) {
            if (!(t->metrics_mask & kd.bit)) continue;
            const uint64_t ttp = joint[kd.base], tfn = joint[kd.base + 1], qtp = joint[kd.base + 2], qfp = joint[kd.base + 3];
            const uint64_t ttot = ttp + tfn, qtot = qtp + qfp;
            row = std::to_string(b->region_id ? b->region_id[r] : r) + "\t" + coords + "\t" + kd.name + "\t" + std::to_string(ttot) + "\t" + std::to_string(ttp) + "\t" +
                  std::to_string(tfn) + "\t" + std::to_string(qtot) + "\t" + std::to_string(qtp) + "\t" + std::to_string(qfp) + "\t";
            if (ttot) row += avf_fmt_f64_((double)ttp / (double)ttot);
            row += "\t";
            if (qtot) row += avf_fmt_f64_((double)qtp / (double)qtot);
            row += "\t";
            if (ttot && qtot) {
                const double rc = (double)ttp / (double)ttot, pr = (double)qtp / (double)qtot;
                row += avf_fmt_f64_(2.0 * rc * pr / (rc + pr));
            }
            row += "\t";
            if (kd.gt) row += std::to_string(joint[AVK_F_GT_TRUTH_FN_GT]) + "\t" + std::to_string(joint[AVK_F_GT_QUERY_FP_GT]);
            else row += "\t";
            row += "\n";
            t->ok = t->ok && t->out->write(row);
        }
    }
    return t->ok ? 0 : avf_fail_(AVK_E_ARG, "write error");
}

/* rows of regions [first, first + n); seq_bytes / seq_len / seq_off / seq_stride are the sequence arrays of the avk_result_batch
 * of exactly these n regions (slot k of region i at seq_off[i] + k * seq_stride[i], lengths seq_len[5 i + k]) */
int avf_region_sequences_rows(avf_table *t, const avf_genome *g, const avk_region_batch *b, uint64_t first, uint64_t n, const int32_t *status,
                              const uint8_t *seq_bytes, const uint32_t *seq_len, const uint64_t *seq_off, const uint32_t *seq_stride) {
    if (!t || t->kind != 1 || !g || !b || !status || !seq_bytes || !seq_len || !seq_off || !seq_stride || first + n > b->n_regions)
        return avf_fail_(AVK_E_ARG, "null or invalid argument");
    std::string row;
    for (uint64_t k = 0; k < n && t->ok; ++k) {
        if (status[k] != 0) continue;
        const uint64_t r = first + k;
        row = std::to_string(b->region_id ? b->region_id[r] : r) + "\t" + avf_genome_name(g, b->contig_idx ? b->contig_idx[r] : 0) + ":" +
              std::to_string(b->start[r] + 1) + "-" + std::to_string(b->end[r]);
        for (int s = 0; s < 5; ++s) {
            row += "\t";
            row.append((const char *)seq_bytes + seq_off[k] + (uint64_t)s * seq_stride[k], seq_len[5 * k + s]);
        }
        row += "\n";
        t->ok = t->ok && t->out->write(row);
    }
    return t->ok ? 0 : avf_fail_(AVK_E_ARG, "write error");
}

int avf_table_close(avf_table *t) {
    if (!t) return 0;
    bool ok = t->ok && t->out->finish();
    ok = (fclose(t->fp) == 0) && ok;
    delete t->out;
    delete t;
    return ok ? 0 : avf_fail_(AVK_E_ARG, "write error while closing a debug table");
}

} /* extern "C" */
