/*
 * avf_mergeout.cpp — the outputs of `aardvark merge`: passing.vcf.gz, regions.bed.gz, failed_regions.bed.gz with their tabix
 * indexes (src/writers/variant_merger.rs:47-349) and the merge summary table (src/writers/merge_summary.rs).
 * Part of libaardvark_feeder.so; host code only.
 */
#include "../../../include/aardvark_feeder.h"

#include "avf_tbx.h"

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

int avf_fail_(int code, const char *fmt, ...); /* avf_feeder.cpp */
std::string avf_csv_field_(const std::string &s, char delim);

namespace {

/* MergeClassification::simplify (src/data_types/merge_benchmark.rs:44-53) */
const char *simplify(uint8_t classification) {
    switch (classification) {
    case AVK_MERGE_DIFFERENT: return "different";
    case AVK_MERGE_NO_CONFLICT: return "no_conflict";
    case AVK_MERGE_MAJORITY_AGREE: return "majority";
    case AVK_MERGE_CONFLICT_SELECTION: return "conflict_select";
    case AVK_MERGE_IDENTICAL: return "identical";
    }
    return nullptr;
}

/* the input indices the classification names: the set bits for no_conflict / majority, the index itself for a selection */
std::vector<uint32_t> member_indices(uint8_t classification, uint64_t members) {
    std::vector<uint32_t> idx;
    if (classification == AVK_MERGE_NO_CONFLICT || classification == AVK_MERGE_MAJORITY_AGREE) {
        for (uint32_t i = 0; i < 64; ++i)
            if (members >> i & 1) idx.push_back(i);
    } else if (classification == AVK_MERGE_CONFLICT_SELECTION) idx.push_back((uint32_t)members);
    return idx;
}

bool make_dirs(const std::string &path) { /* create_dir_all */
    for (size_t at = 1; at <= path.size(); ++at) {
        if (at != path.size() && path[at] != '/') continue;
        const std::string part = path.substr(0, at);
        if (mkdir(part.c_str(), 0777) != 0 && errno != EEXIST) return false;
    }
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

/* meta lines and first sample name of a VCF header */
bool read_header(const char *path, std::vector<std::string> &meta, std::string &first_sample) {
    gzFile in = gzopen(path, "rb");
    if (!in) return false;
    std::string line;
    char chunk[1 << 16];
    bool done = false;
    while (!done && gzgets(in, chunk, sizeof(chunk))) {
        line += chunk;
        if (line.empty() || line.back() != '\n') continue; /* a longer line: keep reading */
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.compare(0, 2, "##") == 0) meta.push_back(line);
        else {
            if (line.compare(0, 6, "#CHROM") == 0) {
                size_t tabs = 0, at = 0;
                while (tabs < 9 && (at = line.find('\t', at)) != std::string::npos) {
                    ++tabs;
                    ++at;
                }
                if (tabs == 9) {
                    const size_t e = line.find('\t', at);
                    first_sample = line.substr(at, e == std::string::npos ? std::string::npos : e - at);
                }
            }
            done = true;
        }
        line.clear();
    }
    gzclose(in);
    return true;
}

int check_results(const avk_multi_batch *b, const uint8_t *classification, const uint64_t *members, const int32_t *status) {
    if (b->n_inputs == 0 || b->n_inputs > 64) return avf_fail_(AVK_E_ARG, "n_inputs must be 1..64, got %u", b->n_inputs);
    for (uint64_t r = 0; r < b->n_regions; ++r) {
        if (status[r] != 0) continue;
        if (!simplify(classification[r])) return avf_fail_(AVK_E_ARG, "region %llu has the unknown classification %u", (unsigned long long)r, classification[r]);
        const std::vector<uint32_t> idx = member_indices(classification[r], members[r]);
        if (classification[r] != AVK_MERGE_DIFFERENT && classification[r] != AVK_MERGE_IDENTICAL && idx.empty())
            return avf_fail_(AVK_E_ARG, "region %llu: classification %s without member inputs", (unsigned long long)r, simplify(classification[r]));
        for (uint32_t i : idx)
            if (i >= b->n_inputs) return avf_fail_(AVK_E_ARG, "region %llu names input %u of %u", (unsigned long long)r, i, b->n_inputs);
    }
    return 0;
}

} // namespace

extern "C" int avf_write_merge_outputs(const char *out_folder, const char *primary_vcf, const char *sample_name, const char *version,
                                       const char *command_line, const avf_genome *g, const avk_multi_batch *b, const char *const *tags,
                                       const int32_t *status, const uint8_t *classification, const uint64_t *members) {
    if (!out_folder || !primary_vcf || !g || !b || !tags || !status || !classification || !members) return avf_fail_(AVK_E_ARG, "null argument");
    int rc = check_results(b, classification, members, status);
    if (rc) return rc;
    for (uint32_t i = 0; i < b->n_inputs; ++i)
        if (!tags[i]) return avf_fail_(AVK_E_ARG, "null tag for input %u", i);
    std::vector<std::string> meta;
    std::string first_sample;
    if (!read_header(primary_vcf, meta, first_sample)) return avf_fail_(AVK_E_ARG, "Error while opening %s", primary_vcf);
    if (!make_dirs(out_folder)) return avf_fail_(AVK_E_ARG, "Error while creating output VCF folder %s", out_folder);
    const std::string sample = sample_name && *sample_name ? sample_name : first_sample;

    avf_tbx::IndexedText vcf, passing_bed, failed_bed;
    /* what the reference adds (variant_merger.rs:85-121), in the layout its VCF library writes a header in */
    const std::vector<avf_tbx::HeaderDef> defs = {
        {"INFO", "SOURCES", "##INFO=<ID=SOURCES,Number=.,Type=String,Description=\"List of tools or technologies that called the same record\">"},
        {"INFO", "MR", "##INFO=<ID=MR,Number=1,Type=String,Description=\"The reason this record was allowed in the merge\">"},
        {"FORMAT", "RI", "##FORMAT=<ID=RI,Number=1,Type=Integer,Description=\"Region ID for the comparison\">"}};
    const std::vector<std::pair<std::string, std::string>> others = {{"aardvark_version", std::string("\"") + (version ? version : "") + "\""},
                                                                      {"aardvark_command", std::string("\"") + (command_line ? command_line : "") + "\""}};
    for (const std::string &m : avf_tbx::vcf_header_lines(meta, defs, others)) vcf.header(m + "\n");
    vcf.header("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample + "\n");

    static const char *const gts[6] = {".", "0/0", "0/1", "0|1", "1|0", "1/1"};
    const uint32_t n_contigs = avf_genome_n_contigs(g), k = b->n_inputs;
    std::string all_tags; /* identical: every input is a source (:135-143) */
    for (uint32_t i = 0; i < k; ++i) {
        if (i) all_tags += ',';
        all_tags += tags[i];
    }
    for (uint64_t r = 0; r < b->n_regions; ++r) {
        const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
        if (status[r] == 0 && c >= n_contigs) return avf_fail_(AVK_E_ARG, "region %llu refers to contig %u of %u", (unsigned long long)r, c, n_contigs);
    }
    auto name_of = [&](uint32_t c) { return std::string(avf_genome_name(g, c)); };
    /* the three files are formatted, compressed and indexed side by side */
    const std::string folder = out_folder;
    bool ok_file[3] = {true, true, true};
    auto bed_file = [&](int failed) {
        avf_tbx::IndexedText &dst = failed ? failed_bed : passing_bed;
        /* write_region (:293-311): 0-based start, end, "<reason>_<region id>"; passing or failed file */
        auto format = [&](uint64_t first, uint64_t last, std::string &text, std::vector<avf_tbx::LineMeta> &lines) {
            for (uint64_t r = first; r < last; ++r) {
                if (status[r] != 0 || (classification[r] == AVK_MERGE_DIFFERENT) != (failed == 1)) continue;
                const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
                const size_t at = text.size();
                text += avf_genome_name(g, c);
                text += '\t';
                text += std::to_string(b->start[r]);
                text += '\t';
                text += std::to_string(b->end[r]);
                text += '\t';
                text += simplify(classification[r]);
                text += '_';
                text += std::to_string(b->region_id[r]);
                text += '\n';
                lines.push_back(avf_tbx::LineMeta{c, (uint32_t)(text.size() - at), (int64_t)b->start[r], (int64_t)b->end[r]});
            }
            return true;
        };
        ok_file[1 + failed] = avf_tbx::format_parallel(b->n_regions, format, name_of, dst) &&
                              dst.finish(folder + (failed ? "/failed_regions.bed.gz" : "/regions.bed.gz"), 0x10000);
    };
    std::thread th_pass(bed_file, 0), th_fail(bed_file, 1);
    auto format_vcf = [&](uint64_t first, uint64_t last, std::string &text, std::vector<avf_tbx::LineMeta> &lines) {
        std::string info;
        for (uint64_t r = first; r < last; ++r) {
            if (status[r] != 0 || classification[r] == AVK_MERGE_DIFFERENT) continue;
            const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
            const char *chrom = avf_genome_name(g, c);
            const uint8_t cls = classification[r];
            const std::vector<uint32_t> idx = member_indices(cls, members[r]);
            const uint32_t source = cls == AVK_MERGE_IDENTICAL ? 0 : idx[0];
            info = "SOURCES=";
            if (cls == AVK_MERGE_IDENTICAL) info += all_tags;
            else
                for (size_t i = 0; i < idx.size(); ++i) {
                    if (i) info += ',';
                    info += tags[idx[i]];
                }
            info += ";MR=";
            info += simplify(cls);
            const std::string ri = std::to_string((int32_t)b->region_id[r]); /* `region_id as i32` (:263) */
            const uint64_t off = b->in_off[r * k + source];
            const uint32_t cnt = b->in_cnt[r * k + source];
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t v = off + i;
                const size_t at = text.size();
                text += chrom;
                text += '\t';
                text += std::to_string(b->var_pos[v] + 1);
                text += "\t.\t";
                text.append((const char *)b->allele_bytes + b->a0_off[v], b->a0_len[v]);
                text += '\t';
                text.append((const char *)b->allele_bytes + b->a1_off[v], b->a1_len[v]);
                text += "\t.\t.\t";
                text += info;
                text += "\tGT:RI\t";
                text += gts[b->var_zyg[v] < 6 ? b->var_zyg[v] : 0];
                text += ':';
                text += ri;
                text += '\n';
                lines.push_back(avf_tbx::LineMeta{c, (uint32_t)(text.size() - at), (int64_t)b->var_pos[v], (int64_t)b->var_pos[v] + (int64_t)(b->a0_len[v] ? b->a0_len[v] : 1)});
            }
        }
        return true;
    };
    ok_file[0] = avf_tbx::format_parallel(b->n_regions, format_vcf, name_of, vcf) && vcf.finish(folder + "/passing.vcf.gz", 2);
    th_pass.join();
    th_fail.join();
    if (!ok_file[0]) return avf_fail_(AVK_E_ARG, "write error on %s/passing.vcf.gz (or its .tbi)", out_folder);
    if (!ok_file[1]) return avf_fail_(AVK_E_ARG, "write error on %s/regions.bed.gz (or its .tbi)", out_folder);
    if (!ok_file[2]) return avf_fail_(AVK_E_ARG, "write error on %s/failed_regions.bed.gz (or its .tbi)", out_folder);
    return 0;
}

namespace {
/* derive(Ord) of MergeClassification: Different < NoConflict{indices} < MajorityAgree{indices} < ConflictSelection{index} <
 * BasepairIdentical, index lists compared lexicographically; then VariantType in declaration order; then the input */
typedef std::tuple<int, std::vector<uint32_t>, uint8_t, uint32_t> SummaryKey;
typedef std::map<SummaryKey, std::pair<uint64_t, uint64_t>> SummaryCounts;
const int summary_rank_of[5] = {0, 4, 1, 2, 3}; /* AVK_MERGE_* -> position in the enum */

int write_summary_rows(const char *path, const char *const *tags, const SummaryCounts &counts) {
    static const char *const type_names[AVK_N_VARIANT_TYPES] = {"Snv", "Insertion", "Deletion", "Indel", "SvInsertion", "SvDeletion", "SvDuplication",
                                                                 "SvInversion", "SvBreakend", "TrContraction", "TrExpansion", "Unknown"};
    const std::string p = path;
    const char delim = p.size() >= 4 && p.compare(p.size() - 4, 4, ".csv") == 0 ? ',' : '\t';
    std::string text;
    const char *const columns[6] = {"merge_reason", "variant_type", "vcf_index", "vcf_label", "pass_variants", "fail_variants"};
    for (int i = 0; i < 6; ++i) {
        if (i) text += delim;
        text += columns[i];
    }
    text += '\n';
    static const uint8_t cls_of_rank[5] = {AVK_MERGE_DIFFERENT, AVK_MERGE_NO_CONFLICT, AVK_MERGE_MAJORITY_AGREE, AVK_MERGE_CONFLICT_SELECTION,
                                           AVK_MERGE_IDENTICAL};
    if (counts.empty()) text.clear(); /* the csv writer emits its header with the first row: no rows, empty file */
    for (const auto &kv : counts) {
        /* Display of MergeClassification (:18-39): the reason, then "_<i>" per index */
        std::string reason = simplify(cls_of_rank[std::get<0>(kv.first)]);
        for (uint32_t i : std::get<1>(kv.first)) reason += '_' + std::to_string(i);
        const uint32_t input = std::get<3>(kv.first);
        text += reason;
        text += delim;
        text += type_names[std::get<2>(kv.first)];
        text += delim;
        text += std::to_string(input);
        text += delim;
        text += avf_csv_field_(tags[input] ? tags[input] : "", delim);
        text += delim;
        text += std::to_string(kv.second.first);
        text += delim;
        text += std::to_string(kv.second.second);
        text += '\n';
    }
    FILE *fp = fopen(path, "wb");
    if (!fp) return avf_fail_(AVK_E_ARG, "cannot create %s", path);
    const bool ok = fwrite(text.data(), 1, text.size(), fp) == text.size();
    if (fclose(fp) != 0 || !ok) return avf_fail_(AVK_E_ARG, "write error on %s", path);
    return 0;
}
} // namespace

extern "C" int avf_write_merge_summary(const char *path, const avk_multi_batch *b, const char *const *tags, const int32_t *status,
                                       const uint8_t *classification, const uint64_t *members) {
    if (!path || !b || !tags || !status || !classification || !members) return avf_fail_(AVK_E_ARG, "null argument");
    int rc = check_results(b, classification, members, status);
    if (rc) return rc;
    SummaryCounts counts;
    const uint32_t k = b->n_inputs;
    for (uint64_t r = 0; r < b->n_regions; ++r) {
        if (status[r] != 0) continue;
        const uint8_t cls = classification[r];
        const std::vector<uint32_t> idx = member_indices(cls, members[r]);
        for (uint32_t i = 0; i < k; ++i) {
            const bool passing = cls == AVK_MERGE_IDENTICAL || std::find(idx.begin(), idx.end(), i) != idx.end();
            const uint64_t off = b->in_off[r * k + i];
            for (uint32_t j = 0; j < b->in_cnt[r * k + i]; ++j) {
                const uint8_t vt = b->var_type[off + j];
                if (vt >= AVK_N_VARIANT_TYPES) return avf_fail_(AVK_E_ARG, "variant %llu has the unknown type %u", (unsigned long long)(off + j), vt);
                std::pair<uint64_t, uint64_t> &e = counts[SummaryKey(summary_rank_of[cls], idx, vt, i)];
                (passing ? e.first : e.second) += 1;
            }
        }
    }
    return write_summary_rows(path, tags, counts);
}

/* The same table from the dense block of sums a sharded merge all-reduces (include/aardvark_amd.h: avk_merge_counts / avk_merge_counts_reason): the keys
 * with a count become the rows, in the reference's key order.  The layout is restated here (the feeder library does not link the solver library):
 * reason 0 Different, 1 + mask NoConflict, 1 + 2^k + mask MajorityAgree, 1 + 2 * 2^k + index ConflictSelection, 1 + 2 * 2^k + k BasepairIdentical. */
extern "C" int avf_write_merge_summary_counts(const char *path, uint32_t n_inputs, const char *const *tags, const uint64_t *counts, uint64_t counts_len) {
    if (!path || !tags || !counts) return avf_fail_(AVK_E_ARG, "null argument");
    const uint32_t k = n_inputs;
    if (k < 2 || k > AVK_MERGE_COUNTS_MAX_INPUTS) return avf_fail_(AVK_E_ARG, "dense summary counters exist for 2..%d inputs, not %u", AVK_MERGE_COUNTS_MAX_INPUTS, k);
    const uint64_t masks = 1ull << k, reasons = 2 + 2 * masks + k;
    if (counts_len != reasons * AVK_N_VARIANT_TYPES * k * 2) return avf_fail_(AVK_E_ARG, "%llu counters given, %u inputs have %llu", (unsigned long long)counts_len, k,
                                                                              (unsigned long long)(reasons * AVK_N_VARIANT_TYPES * k * 2));
    SummaryCounts rows;
    for (uint64_t reason = 0; reason < reasons; ++reason) {
        uint8_t cls = AVK_MERGE_DIFFERENT;
        uint64_t members = 0;
        if (reason == reasons - 1) cls = AVK_MERGE_IDENTICAL;
        else if (reason >= 1 + 2 * masks) cls = AVK_MERGE_CONFLICT_SELECTION, members = reason - 1 - 2 * masks;
        else if (reason >= 1 + masks) cls = AVK_MERGE_MAJORITY_AGREE, members = reason - 1 - masks;
        else if (reason >= 1) cls = AVK_MERGE_NO_CONFLICT, members = reason - 1;
        const std::vector<uint32_t> idx = member_indices(cls, members);
        for (uint32_t vt = 0; vt < AVK_N_VARIANT_TYPES; ++vt)
            for (uint32_t i = 0; i < k; ++i) {
                const uint64_t *e = counts + ((reason * AVK_N_VARIANT_TYPES + vt) * k + i) * 2;
                if (e[0] || e[1]) rows[SummaryKey(summary_rank_of[cls], idx, (uint8_t)vt, i)] = std::make_pair(e[0], e[1]);
            }
    }
    return write_summary_rows(path, tags, rows);
}
