/*
 * avf_vcfout.cpp — the annotated truth.vcf.gz / query.vcf.gz of `aardvark compare`
 * (src/writers/variant_categorizer.rs:41-230, src/writers/noodles_idx.rs:7-20), written as BGZF with a tabix index.
 * Part of libaardvark_feeder.so; host code only.
 */
#include "../../../include/aardvark_feeder.h"

#include "avf_tbx.h"

#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

int avf_fail_(int code, const char *fmt, ...); /* avf_feeder.cpp: sets the text avf_last_error returns */

extern "C" int avf_write_annotated_vcf(const char *out_path, const char *input_vcf, const char *sample_name, const char *version, const char *command_line,
                                       const avf_genome *g, const avk_region_batch *b, int source, const int32_t *status, const uint8_t *var_expected,
                                       const uint8_t *var_observed, const uint8_t *var_class) {
    if (!out_path || !input_vcf || !g || !b || !status || !var_expected || !var_observed || !var_class || (source != 0 && source != 1))
        return avf_fail_(AVK_E_ARG, "null or invalid argument");
    /* header of the input file, up to the column line */
    std::vector<std::string> meta;
    std::string first_sample;
    {
        gzFile in = gzopen(input_vcf, "rb");
        if (!in) return avf_fail_(AVK_E_ARG, "Error while opening %s", input_vcf);
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
                    if (tabs == 9) first_sample = line.substr(at, line.find('\t', at) == std::string::npos ? std::string::npos : line.find('\t', at) - at);
                }
                done = true;
            }
            line.clear();
        }
        gzclose(in);
    }
    const std::string sample = sample_name && *sample_name ? sample_name : first_sample;
    avf_tbx::IndexedText out;
    /* what the reference adds (variant_categorizer.rs:41-87), in the layout its VCF library writes a header in */
    const std::vector<avf_tbx::HeaderDef> defs = {
        {"FORMAT", "BD", "##FORMAT=<ID=BD,Number=1,Type=String,Description=\"Benchmark Decision for call (TP/FP/FN)\">"},
        {"FORMAT", "EA", "##FORMAT=<ID=EA,Number=1,Type=Integer,Description=\"Expected Allele count for this genotype\">"},
        {"FORMAT", "OA", "##FORMAT=<ID=OA,Number=1,Type=Integer,Description=\"Observed Allele count for this genotype\">"},
        {"FORMAT", "RI", "##FORMAT=<ID=RI,Number=1,Type=Integer,Description=\"Region ID for the comparison\">"}};
    const std::vector<std::pair<std::string, std::string>> others = {{"aardvark_version", std::string("\"") + (version ? version : "") + "\""},
                                                                      {"aardvark_command", std::string("\"") + (command_line ? command_line : "") + "\""}};
    for (const std::string &m : avf_tbx::vcf_header_lines(meta, defs, others)) out.header(m + "\n");
    out.header("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample + "\n");

    static const char *const gts[6] = {".", "0/0", "0/1", "0|1", "1|0", "1/1"};
    static const char *const classes[4] = {"UNK", "TP", "FN", "FP"};
    const uint32_t n_contigs = avf_genome_n_contigs(g);
    for (uint64_t r = 0; r < b->n_regions; ++r) {
        const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
        if (status[r] == 0 && c >= n_contigs) return avf_fail_(AVK_E_ARG, "region %llu refers to contig %u of %u", (unsigned long long)r, c, n_contigs);
    }
    auto format = [&](uint64_t first, uint64_t last, std::string &text, std::vector<avf_tbx::LineMeta> &lines) {
        char num[24];
        auto put_u64 = [&](uint64_t v) { text.append(num, (size_t)(std::to_chars(num, num + sizeof(num), v).ptr - num)); };
        auto put_i32 = [&](int32_t v) { text.append(num, (size_t)(std::to_chars(num, num + sizeof(num), v).ptr - num)); };
        {
            uint64_t n_lines = 0, n_allele = 0;
            for (uint64_t r = first; r < last; ++r) {
                if (status[r] != 0) continue;
                const uint64_t off = source == 0 ? b->t_off[r] : b->q_off[r];
                const uint32_t cnt = source == 0 ? b->t_cnt[r] : b->q_cnt[r];
                n_lines += cnt;
                for (uint32_t i = 0; i < cnt; ++i) n_allele += (uint64_t)b->a0_len[off + i] + b->a1_len[off + i];
            }
            text.reserve(text.size() + n_lines * 64 + n_allele);
            lines.reserve(lines.size() + n_lines);
        }
        for (uint64_t r = first; r < last; ++r) {
            if (status[r] != 0) continue; /* failed regions are not written (compare_parallel.rs:229-262) */
            const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
            const char *chrom = avf_genome_name(g, c);
            const size_t chrom_len = strlen(chrom);
            const uint64_t off = source == 0 ? b->t_off[r] : b->q_off[r];
            const uint32_t cnt = source == 0 ? b->t_cnt[r] : b->q_cnt[r];
            char ri[16];
            const size_t ri_len = (size_t)(std::to_chars(ri, ri + sizeof(ri), (int32_t)b->region_id[r]).ptr - ri); /* `region_id as i32` (:205) */
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t v = off + i;
                const size_t at = text.size();
                text.append(chrom, chrom_len);
                text += '\t';
                put_u64(b->var_pos[v] + 1);
                text += "\t.\t";
                text.append((const char *)b->allele_bytes + b->a0_off[v], b->a0_len[v]);
                text += '\t';
                text.append((const char *)b->allele_bytes + b->a1_off[v], b->a1_len[v]);
                text += "\t.\t.\t.\tGT:BD:EA:OA:RI\t";
                text += gts[b->var_zyg[v] < 6 ? b->var_zyg[v] : 0];
                text += ':';
                text += classes[var_class[v] < 4 ? var_class[v] : 0];
                text += ':';
                put_i32((int32_t)var_expected[v]);
                text += ':';
                put_i32((int32_t)var_observed[v]);
                text += ':';
                text.append(ri, ri_len);
                text += '\n';
                /* index entry: [beg, end) = POS-1 .. POS-1 + len(REF) */
                lines.push_back(avf_tbx::LineMeta{c, (uint32_t)(text.size() - at), (int64_t)b->var_pos[v], (int64_t)b->var_pos[v] + (int64_t)(b->a0_len[v] ? b->a0_len[v] : 1)});
            }
        }
        return true;
    };
    if (!avf_tbx::format_parallel(b->n_regions, format, [&](uint32_t c) { return std::string(avf_genome_name(g, c)); }, out))
        return avf_fail_(AVK_E_ARG, "cannot format the records of %s", out_path);
    if (!out.finish(out_path, 2)) return avf_fail_(AVK_E_ARG, "write error on %s (or its .tbi)", out_path);
    return 0;
}
