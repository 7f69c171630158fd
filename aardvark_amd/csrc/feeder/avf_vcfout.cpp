/*
 * avf_vcfout.cpp — the annotated truth.vcf.gz / query.vcf.gz of `aardvark compare`
 * (src/writers/variant_categorizer.rs:41-230, src/writers/noodles_idx.rs:7-20), written as BGZF with a tabix index.
 * Part of libaardvark_feeder.so; host code only.
 */
#include "../../../include/aardvark_feeder.h"

#include "avf_bgzf.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

int avf_fail_(int code, const char *fmt, ...); /* avf_feeder.cpp: sets the text avf_last_error returns */

namespace {

/* ---- BGZF (SAM spec section 4.1): gzip members of at most 64 KiB with a BC extra field ----
 * The text is cut into blocks as it is written; the blocks are compressed by a few threads at the end (like the
 * reference's MultithreadedWriter, variant_categorizer.rs:108-110).  tell() therefore returns a LOGICAL virtual offset
 * (block index << 16 | offset in block); real() turns it into the file's virtual offset once the block sizes are known. */
class BgzfWriter {
  public:
    BgzfWriter() { blocks_.emplace_back(); }
    uint64_t tell() const { return ((uint64_t)(blocks_.size() - 1) << 16) | (uint64_t)blocks_.back().size(); }
    void write(const char *p, size_t n) {
        while (n) {
            std::string &cur = blocks_.back();
            const size_t room = kBlock - cur.size();
            const size_t take = n < room ? n : room;
            cur.append(p, take);
            p += take;
            n -= take;
            if (blocks_.back().size() == kBlock) blocks_.emplace_back();
        }
    }
    /* compresses and writes everything plus the end-of-file block */
    bool finish(FILE *fp, int threads) {
        if (blocks_.back().empty()) blocks_.pop_back();
        const size_t nb = blocks_.size();
        std::vector<std::string> packed(nb);
        std::vector<char> bad(nb, 0);
        int nt = threads < 1 ? 1 : threads;
        if ((size_t)nt > nb) nt = nb ? (int)nb : 1;
        auto work = [&](size_t t) {
            for (size_t k = t; k < nb; k += (size_t)nt)
                if (!compress_block(blocks_[k], packed[k])) bad[k] = 1;
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, (size_t)t);
        work(0);
        for (auto &th : pool) th.join();
        file_off_.assign(nb + 1, 0);
        for (size_t k = 0; k < nb; ++k) {
            if (bad[k]) return false;
            file_off_[k + 1] = file_off_[k] + packed[k].size();
            if (fwrite(packed[k].data(), 1, packed[k].size(), fp) != packed[k].size()) return false;
        }
        return avf_bgzf::write_eof(fp);
    }
    /* after finish(): the file's virtual offset of a logical one (an offset at the very end of a block is the start of the next) */
    uint64_t real(uint64_t logical) const {
        size_t blk = (size_t)(logical >> 16);
        uint64_t off = logical & 0xFFFF;
        if (blk >= file_off_.size() - 1) return file_off_.back() << 16;
        return (file_off_[blk] << 16) | off;
    }

  private:
    static constexpr size_t kBlock = avf_bgzf::kBlock;
    static bool compress_block(const std::string &in, std::string &out) { return avf_bgzf::compress_block(in.data(), in.size(), out); }
    std::vector<std::string> blocks_;
    std::vector<uint64_t> file_off_;
};

/* ---- tabix index (tabix spec): binning index + 16 kb linear index per contig ---- */
int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

const uint64_t kNone = ~0ull;

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
};

template <typename T> void put(std::string &s, T v) {
    for (size_t k = 0; k < sizeof(T); ++k) s.push_back((char)((uint64_t)v >> (8 * k)));
}

} // namespace

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
    FILE *fp = fopen(out_path, "wb");
    if (!fp) return avf_fail_(AVK_E_ARG, "cannot create %s", out_path);
    BgzfWriter w;
    bool ok = true;
    auto emit = [&](const std::string &s) { w.write(s.data(), s.size()); };
    for (const std::string &m : meta) emit(m + "\n");
    /* what the reference adds (variant_categorizer.rs:41-87) */
    emit(std::string("##aardvark_version=\"") + (version ? version : "") + "\"\n");
    emit(std::string("##aardvark_command=\"") + (command_line ? command_line : "") + "\"\n");
    emit("##FORMAT=<ID=BD,Number=1,Type=String,Description=\"Benchmark Decision for call (TP/FP/FN)\">\n");
    emit("##FORMAT=<ID=EA,Number=1,Type=Integer,Description=\"Expected Allele count for this genotype\">\n");
    emit("##FORMAT=<ID=OA,Number=1,Type=Integer,Description=\"Observed Allele count for this genotype\">\n");
    emit("##FORMAT=<ID=RI,Number=1,Type=Integer,Description=\"Region ID for the comparison\">\n");
    emit("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample + "\n");

    static const char *const gts[6] = {".", "0/0", "0/1", "0|1", "1|0", "1/1"};
    static const char *const classes[4] = {"UNK", "TP", "FN", "FP"};
    const uint32_t n_contigs = avf_genome_n_contigs(g);
    std::vector<RefIndex> index(n_contigs);
    std::vector<uint32_t> contig_order; /* contigs in order of first record */
    std::vector<char> seen(n_contigs, 0);
    std::string rec;
    for (uint64_t r = 0; r < b->n_regions && ok; ++r) {
        if (status[r] != 0) continue; /* failed regions are not written (compare_parallel.rs:229-262) */
        const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
        if (c >= n_contigs) {
            fclose(fp);
            return avf_fail_(AVK_E_ARG, "region %llu refers to contig %u of %u", (unsigned long long)r, c, n_contigs);
        }
        const uint64_t off = source == 0 ? b->t_off[r] : b->q_off[r];
        const uint32_t cnt = source == 0 ? b->t_cnt[r] : b->q_cnt[r];
        for (uint32_t i = 0; i < cnt && ok; ++i) {
            const uint64_t v = off + i;
            rec.assign(avf_genome_name(g, c));
            rec += '\t';
            rec += std::to_string(b->var_pos[v] + 1);
            rec += "\t.\t";
            rec.append((const char *)b->allele_bytes + b->a0_off[v], b->a0_len[v]);
            rec += '\t';
            rec.append((const char *)b->allele_bytes + b->a1_off[v], b->a1_len[v]);
            rec += "\t.\t.\t.\tGT:BD:EA:OA:RI\t";
            rec += gts[b->var_zyg[v] < 6 ? b->var_zyg[v] : 0];
            rec += ':';
            rec += classes[var_class[v] < 4 ? var_class[v] : 0];
            rec += ':';
            rec += std::to_string((int)var_expected[v]);
            rec += ':';
            rec += std::to_string((int)var_observed[v]);
            rec += ':';
            rec += std::to_string((int32_t)b->region_id[r]); /* `region_id as i32` (:205) */
            rec += '\n';
            const uint64_t vbeg = w.tell();
            emit(rec);
            const uint64_t vend = w.tell();
            /* index entry: [beg, end) = POS-1 .. POS-1 + len(REF) */
            if (!seen[c]) {
                seen[c] = 1;
                contig_order.push_back(c);
            }
            const int64_t beg = (int64_t)b->var_pos[v], end = beg + (int64_t)(b->a0_len[v] ? b->a0_len[v] : 1);
            RefIndex &ri = index[c];
            auto &chunks = ri.bins[(uint32_t)reg2bin(beg, end)];
            if (!chunks.empty() && chunks.back().second == vbeg) chunks.back().second = vend;
            else chunks.emplace_back(vbeg, vend);
            const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
            if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, kNone);
            for (size_t k = w0; k <= w1; ++k)
                if (ri.linear[k] == kNone) ri.linear[k] = vbeg;
        }
    }
    int threads = (int)std::thread::hardware_concurrency();
    if (threads > 8) threads = 8;
    ok = ok && w.finish(fp, threads);
    if (fclose(fp) != 0 || !ok) return avf_fail_(AVK_E_ARG, "write error on %s", out_path);
    /* logical virtual offsets -> the file's */
    for (RefIndex &ri : index) {
        for (auto &kv : ri.bins)
            for (auto &ch : kv.second) {
                ch.first = w.real(ch.first);
                ch.second = w.real(ch.second);
            }
        for (uint64_t &o : ri.linear)
            if (o != kNone) o = w.real(o);
    }

    /* the .tbi next to it */
    std::string tbi;
    tbi.append("TBI\1", 4);
    put<int32_t>(tbi, (int32_t)contig_order.size());
    put<int32_t>(tbi, 2);   /* format: VCF */
    put<int32_t>(tbi, 1);   /* col_seq */
    put<int32_t>(tbi, 2);   /* col_beg */
    put<int32_t>(tbi, 0);   /* col_end */
    put<int32_t>(tbi, '#'); /* meta */
    put<int32_t>(tbi, 0);   /* skip */
    std::string names;
    for (uint32_t c : contig_order) {
        names += avf_genome_name(g, c);
        names.push_back('\0');
    }
    put<int32_t>(tbi, (int32_t)names.size());
    tbi += names;
    for (uint32_t c : contig_order) {
        RefIndex &ri = index[c];
        put<int32_t>(tbi, (int32_t)ri.bins.size());
        for (auto &kv : ri.bins) {
            put<uint32_t>(tbi, kv.first);
            put<int32_t>(tbi, (int32_t)kv.second.size());
            for (auto &ch : kv.second) {
                put<uint64_t>(tbi, ch.first);
                put<uint64_t>(tbi, ch.second);
            }
        }
        /* windows without a record inherit the offset of the next one before them (htslib convention) */
        for (size_t k = 0; k < ri.linear.size(); ++k)
            if (ri.linear[k] == kNone) ri.linear[k] = k ? ri.linear[k - 1] : 0;
        put<int32_t>(tbi, (int32_t)ri.linear.size());
        for (uint64_t o : ri.linear) put<uint64_t>(tbi, o);
    }
    const std::string tbi_path = std::string(out_path) + ".tbi";
    FILE *tf = fopen(tbi_path.c_str(), "wb");
    if (!tf) return avf_fail_(AVK_E_ARG, "cannot create %s", tbi_path.c_str());
    BgzfWriter tw;
    tw.write(tbi.data(), tbi.size());
    const bool tok = tw.finish(tf, 1);
    if (fclose(tf) != 0 || !tok) return avf_fail_(AVK_E_ARG, "write error on %s", tbi_path.c_str());
    return 0;
}
