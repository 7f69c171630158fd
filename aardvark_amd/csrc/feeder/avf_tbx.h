/* avf_tbx.h — a BGZF text file with a tabix index next to it (VCF or BED records), for the writers of libaardvark_feeder.so */
#ifndef AVF_TBX_H
#define AVF_TBX_H
#include "../avk_cpus.h"
#include "avf_bgzf.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace avf_tbx {

/* ---- BGZF (SAM spec section 4.1): gzip members of at most 64 KiB with a BC extra field ----
 * The text is cut into blocks as it is written; the blocks are compressed by a few threads at the end (like the
 * reference's MultithreadedWriter, variant_categorizer.rs:108-110).  tell() therefore returns a LOGICAL virtual offset
 * (block index << 16 | offset in block); real() turns it into the file's virtual offset once the block sizes are known. */
class BgzfWriter {
  public:
    BgzfWriter() { blocks_.emplace_back(); }
    uint64_t tell() const { return ((uint64_t)(blocks_.size() - 1) << 16) | (uint64_t)blocks_.back().size(); }
    /* bytes written so far, and the logical virtual offset of byte `abs` of the text (every block but the last one is full) */
    uint64_t size() const { return total_; }
    static uint64_t logical_at(uint64_t abs) { return ((abs / kBlock) << 16) | (abs % kBlock); }
    void write(const char *p, size_t n) {
        total_ += n;
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
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, (size_t)t);
        work(0);
        for (auto &th : pool) th.join();
        if (getenv("AVF_TIMING"))
            fprintf(stderr, "[avf] bgzf: %zu blocks compressed by %d threads in %.3f s\n", nb, nt, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
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
    uint64_t total_ = 0;
};

/* ---- tabix index (tabix spec): binning index + 16 kb linear index per contig ---- */
inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

constexpr uint64_t kNone = ~0ull;

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
};

template <typename T> inline void put(std::string &s, T v) {
    for (size_t k = 0; k < sizeof(T); ++k) s.push_back((char)((uint64_t)v >> (8 * k)));
}


/* Collects the text and the index entries of one output file; finish() compresses, writes path and path + ".tbi".
 * Records must arrive sorted by position within a sequence; sequences are indexed in order of first appearance. */
class IndexedText {
  public:
    void header(const std::string &s) { w_.write(s.data(), s.size()); }
    /* one record line covering [beg, end) (0-based) of sequence `name` */
    void record(const std::string &name, int64_t beg, int64_t end, const std::string &line) {
        const size_t k = seq_id(name);
        const uint64_t at = w_.size();
        w_.write(line.data(), line.size());
        index_record(k, beg, end, at, at + line.size());
    }
    /* the same in two steps, for text formatted elsewhere (by several threads): the sequence's id, then append() the text and
     * index_record() every line of it with its byte range in the whole text */
    size_t seq_id(const std::string &name) {
        if (last_seq_ < names_.size() && names_[last_seq_] == name) return last_seq_;
        auto it = seq_of_.find(name);
        if (it == seq_of_.end()) {
            last_seq_ = names_.size();
            seq_of_.emplace(name, last_seq_);
            names_.push_back(name);
            index_.emplace_back();
            last_chunks_ = nullptr;
        } else last_seq_ = it->second;
        return last_seq_;
    }
    uint64_t text_size() const { return w_.size(); }
    void append(const char *p, size_t n) { w_.write(p, n); }
    void index_record(size_t k, int64_t beg, int64_t end, uint64_t abs_beg, uint64_t abs_end) {
        const uint64_t vbeg = BgzfWriter::logical_at(abs_beg), vend = BgzfWriter::logical_at(abs_end);
        if (end <= beg) end = beg + 1;
        RefIndex &ri = index_[k];
        const uint32_t bin = (uint32_t)reg2bin(beg, end);
        if (!last_chunks_ || last_bin_seq_ != k || last_bin_ != bin) { /* neighbours mostly share a bin */
            last_chunks_ = &ri.bins[bin];
            last_bin_seq_ = k;
            last_bin_ = bin;
        }
        auto &chunks = *last_chunks_;
        if (!chunks.empty() && chunks.back().second == vbeg) chunks.back().second = vend;
        else chunks.emplace_back(vbeg, vend);
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, kNone);
        for (size_t q = w0; q <= w1; ++q)
            if (ri.linear[q] == kNone) ri.linear[q] = vbeg;
    }
    /* format: 2 = VCF (col_seq 1, col_beg 2, col_end 0), 0x10000 = zero-based BED (1, 2, 3) */
    bool finish(const std::string &path, int format) {
        const bool timing = getenv("AVF_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        FILE *fp = fopen(path.c_str(), "wb");
        if (!fp) return false;
        int threads = (int)avk_usable_cpus();
        if (threads > 64) threads = 64;
        bool ok = w_.finish(fp, threads);
        ok = (fclose(fp) == 0) && ok;
        if (!ok) return false;
        if (timing) fprintf(stderr, "[avf] compress + write %s: %.3f s\n", path.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        for (RefIndex &ri : index_) { /* logical virtual offsets -> the file's */
            for (auto &kv : ri.bins)
                for (auto &ch : kv.second) {
                    ch.first = w_.real(ch.first);
                    ch.second = w_.real(ch.second);
                }
            for (uint64_t &o : ri.linear)
                if (o != kNone) o = w_.real(o);
        }
        std::string tbi;
        tbi.append("TBI\1", 4);
        put<int32_t>(tbi, (int32_t)names_.size());
        put<int32_t>(tbi, format);
        put<int32_t>(tbi, 1);
        put<int32_t>(tbi, 2);
        put<int32_t>(tbi, format == 2 ? 0 : 3);
        put<int32_t>(tbi, '#');
        put<int32_t>(tbi, 0);
        std::string names;
        for (const std::string &n : names_) {
            names += n;
            names.push_back('\0');
        }
        put<int32_t>(tbi, (int32_t)names.size());
        tbi += names;
        for (RefIndex &ri : index_) {
            put<int32_t>(tbi, (int32_t)ri.bins.size());
            for (auto &kv : ri.bins) {
                put<uint32_t>(tbi, kv.first);
                put<int32_t>(tbi, (int32_t)kv.second.size());
                for (auto &ch : kv.second) {
                    put<uint64_t>(tbi, ch.first);
                    put<uint64_t>(tbi, ch.second);
                }
            }
            /* windows without a record inherit the offset of the one before them (htslib convention) */
            for (size_t q = 0; q < ri.linear.size(); ++q)
                if (ri.linear[q] == kNone) ri.linear[q] = q ? ri.linear[q - 1] : 0;
            put<int32_t>(tbi, (int32_t)ri.linear.size());
            for (uint64_t o : ri.linear) put<uint64_t>(tbi, o);
        }
        FILE *tf = fopen((path + ".tbi").c_str(), "wb");
        if (!tf) return false;
        BgzfWriter tw;
        tw.write(tbi.data(), tbi.size());
        const bool tok = tw.finish(tf, threads); /* a whole-genome index is several megabytes */
        return (fclose(tf) == 0) && tok;
    }

  private:
    BgzfWriter w_;
    std::vector<std::string> names_;
    std::map<std::string, size_t> seq_of_;
    std::vector<RefIndex> index_;
    size_t last_seq_ = (size_t)-1, last_bin_seq_ = (size_t)-1;
    uint32_t last_bin_ = 0;
    std::vector<std::pair<uint64_t, uint64_t>> *last_chunks_ = nullptr; /* std::map nodes do not move */
};

/* One structured meta line, `##KEY=<k=v,...>`, written again field by field the way the reference's VCF library (noodles-vcf 0.80, a Cargo dependency
 * that is not vendored: restated from its published writer, not pinned by a reference run) writes a definition it has parsed:
 *   INFO / FORMAT   ID, Number, Type, Description, then the other fields, then IDX
 *   FILTER / ALT    ID, Description, the other fields, IDX
 *   contig          ID, length, md5, URL, the other fields, IDX
 *   any other key   the first field (its "ID tag"), then the other fields
 * ID, Number, Type, length, md5, URL and IDX bare; Description and every "other" field in double quotes with `\` and `"` escaped, whether or not the input
 * quoted them (`Source=x,Version=3` comes out as `Source="x",Version="3"`).  A line that does not parse (no closing `>`, a field without `=`, a definition
 * without the fields its kind requires — the reference's reader would refuse the file) is returned as it is. */
inline std::string vcf_reserialise_definition(const std::string &line) {
    const size_t eq = line.find('=');
    if (line.compare(0, 2, "##") != 0 || eq == std::string::npos || eq + 1 >= line.size() || line[eq + 1] != '<' || line.back() != '>') return line;
    const std::string key = line.substr(2, eq - 2);
    struct Field {
        std::string k, v;
    };
    std::vector<Field> fields;
    size_t at = eq + 2;
    const size_t end = line.size() - 1;
    while (at < end) {
        const size_t ke = line.find('=', at);
        if (ke == std::string::npos || ke >= end) return line;
        Field f;
        f.k = line.substr(at, ke - at);
        if (f.k.empty() || f.k.find(',') != std::string::npos) return line;
        at = ke + 1;
        if (at < end && line[at] == '"') {
            ++at;
            bool closed = false;
            while (at < end) {
                const char ch = line[at++];
                if (ch == '\\' && at < end && (line[at] == '"' || line[at] == '\\')) f.v.push_back(line[at++]);
                else if (ch == '"') {
                    closed = true;
                    break;
                } else f.v.push_back(ch);
            }
            if (!closed) return line;
            if (at < end && line[at] != ',') return line;
        } else {
            const size_t ve = std::min(line.find(',', at), end);
            f.v = line.substr(at, ve - at);
            at = ve;
        }
        if (at < end) ++at; /* the comma */
        fields.push_back(std::move(f));
    }
    if (fields.empty()) return line;
    auto quoted = [](const std::string &v) {
        std::string q = "\"";
        for (const char ch : v) {
            if (ch == '"' || ch == '\\') q.push_back('\\');
            q.push_back(ch);
        }
        return q + "\"";
    };
    std::vector<bool> used(fields.size(), false);
    std::string out = "##" + key + "=<";
    bool first = true;
    auto take = [&](const char *name, bool quote, bool required) {
        for (size_t i = 0; i < fields.size(); ++i)
            if (!used[i] && fields[i].k == name) {
                used[i] = true;
                if (!first) out += ',';
                first = false;
                out += fields[i].k + "=" + (quote ? quoted(fields[i].v) : fields[i].v);
                return true;
            }
        return !required;
    };
    auto rest = [&](bool with_idx) {
        for (size_t i = 0; i < fields.size(); ++i)
            if (!used[i] && !(with_idx && fields[i].k == "IDX")) {
                used[i] = true;
                out += (first ? "" : ",") + fields[i].k + "=" + quoted(fields[i].v);
                first = false;
            }
        if (with_idx) take("IDX", false, false);
    };
    bool ok = true;
    if (key == "INFO" || key == "FORMAT") {
        ok = take("ID", false, true) && take("Number", false, true) && take("Type", false, true) && take("Description", true, true);
        rest(true);
    } else if (key == "FILTER" || key == "ALT") {
        ok = take("ID", false, true) && take("Description", true, true);
        rest(key == "FILTER");
    } else if (key == "contig") {
        ok = take("ID", false, true);
        take("length", false, false);
        take("md5", false, false);
        take("URL", false, false);
        rest(true);
    } else {
        used[0] = true;
        out += fields[0].k + "=" + fields[0].v;
        first = false;
        rest(false);
    }
    return ok ? out + ">" : line;
}

/* The header of an output VCF the way the reference's VCF library serialises a header it has parsed and extended (noodles: the
 * file format line, then the INFO, FILTER, FORMAT, ALT and contig definitions as groups, then every other line grouped by its key in
 * order of first appearance): the input's meta lines regrouped and its structured lines written again field by field
 * (vcf_reserialise_definition), `defs` added to (or replacing the same ID in) the INFO / FORMAT groups, `others` (key, value) appended
 * under their keys. */
struct HeaderDef {
    const char *group; /* "INFO" or "FORMAT" */
    const char *id;
    const char *line;  /* the whole meta line */
};
inline std::vector<std::string> vcf_header_lines(const std::vector<std::string> &meta, const std::vector<HeaderDef> &defs,
                                                 const std::vector<std::pair<std::string, std::string>> &others) {
    static const char *const kGroups[5] = {"INFO", "FILTER", "FORMAT", "ALT", "contig"};
    std::string fileformat;
    std::vector<std::pair<std::string, std::string>> grouped[5]; /* (ID, line) */
    std::vector<std::pair<std::string, std::vector<std::string>>> rest; /* key -> lines */
    auto id_of = [](const std::string &line) {
        const size_t lt = line.find('<');
        size_t at = lt == std::string::npos ? std::string::npos : line.find("ID=", lt);
        if (at == std::string::npos) return std::string();
        at += 3;
        size_t e = at;
        while (e < line.size() && line[e] != ',' && line[e] != '>') ++e;
        return line.substr(at, e - at);
    };
    auto add_other = [&](const std::string &key, const std::string &line) {
        for (auto &kv : rest)
            if (kv.first == key) {
                kv.second.push_back(line);
                return;
            }
        rest.emplace_back(key, std::vector<std::string>(1, line));
    };
    for (const std::string &m : meta) {
        if (m.compare(0, 2, "##") != 0) continue;
        const size_t eq = m.find('=');
        const std::string key = m.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
        if (key == "fileformat") {
            if (fileformat.empty()) fileformat = m;
            continue;
        }
        int g = -1;
        for (int k = 0; k < 5; ++k)
            if (key == kGroups[k] && eq != std::string::npos && eq + 1 < m.size() && m[eq + 1] == '<') g = k;
        if (g >= 0) grouped[g].emplace_back(id_of(m), vcf_reserialise_definition(m));
        else add_other(key, vcf_reserialise_definition(m)); /* (unstructured lines come back as they are) */
    }
    for (const HeaderDef &d : defs) {
        const int g = strcmp(d.group, "INFO") == 0 ? 0 : 2;
        bool replaced = false;
        for (auto &kv : grouped[g])
            if (kv.first == d.id) {
                kv.second = d.line;
                replaced = true;
            }
        if (!replaced) grouped[g].emplace_back(d.id, d.line);
    }
    for (const auto &kv : others) add_other(kv.first, "##" + kv.first + "=" + kv.second);
    std::vector<std::string> out;
    if (!fileformat.empty()) out.push_back(fileformat);
    for (int k = 0; k < 5; ++k)
        for (const auto &kv : grouped[k]) out.push_back(kv.second);
    for (const auto &kv : rest)
        for (const std::string &l : kv.second) out.push_back(l);
    return out;
}

/* Record lines formatted by several threads: fn(first, last, text, lines) appends the lines of items [first, last) to `text` and one
 * LineMeta per line to `lines` (false = error); the pieces are then appended to `out` and indexed in item order.
 * name_of(contig) gives the sequence name of LineMeta::contig. */
struct LineMeta {
    uint32_t contig;
    uint32_t len; /* bytes of the line, terminator included */
    int64_t beg, end;
};
template <class Fn, class NameOf> bool format_parallel(uint64_t n_items, Fn &&fn, NameOf &&name_of, IndexedText &out) {
    unsigned hw = avk_usable_cpus();
    const size_t n_threads = hw < 1 ? 1 : (hw > 32 ? 32 : hw);
    const size_t n_pieces = n_items < 512 ? 1 : std::min<size_t>(4 * n_threads, (size_t)(n_items / 128));
    std::vector<std::string> texts(n_pieces);
    std::vector<std::vector<LineMeta>> metas(n_pieces);
    std::vector<char> bad(n_pieces, 0);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t k = next.fetch_add(1); k < n_pieces; k = next.fetch_add(1)) {
            const uint64_t first = n_items * k / n_pieces, last = n_items * (k + 1) / n_pieces;
            if (!fn(first, last, texts[k], metas[k])) bad[k] = 1;
        }
    };
    const bool timing = getenv("AVF_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (size_t t = 1; t < std::min(n_threads, n_pieces); ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (timing) fprintf(stderr, "[avf] format %zu pieces: %.3f s\n", n_pieces, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    for (size_t k = 0; k < n_pieces; ++k)
        if (bad[k]) return false;
    for (size_t k = 0; k < n_pieces; ++k) {
        uint64_t at = out.text_size();
        out.append(texts[k].data(), texts[k].size());
        uint32_t last_contig = 0xFFFFFFFFu;
        size_t seq = 0;
        for (const LineMeta &m : metas[k]) {
            if (m.contig != last_contig) {
                seq = out.seq_id(name_of(m.contig));
                last_contig = m.contig;
            }
            out.index_record(seq, m.beg, m.end, at, at + m.len);
            at += m.len;
        }
        std::string().swap(texts[k]);
        std::vector<LineMeta>().swap(metas[k]);
    }
    if (timing) fprintf(stderr, "[avf] format + append + index: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return true;
}

} // namespace avf_tbx
#endif
