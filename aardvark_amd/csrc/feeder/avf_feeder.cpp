/*
 * avf_feeder.cpp — libaardvark_feeder.so: FASTA / BED / VCF readers, the reference's region generation and its
 * summary writer, restated in C++ behind the C-ABI of include/aardvark_feeder.h.  Host code only.
 * file:line citations are into PacificBiosciences/aardvark v0.10.5.
 */
#include "../../../include/aardvark_feeder.h"

#include "../avk_cpus.h"
#include "avf_bgzf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string t_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_error = buf;
    return code;
}

} // namespace

/* for the other translation units of the library */
int avf_fail_(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_error = buf;
    return code;
}

namespace {

/* line reader over plain, gzip and BGZF files (zlib reads all three; BGZF is a series of gzip members) */
class LineReader {
  public:
    explicit LineReader(const char *path) : gz_(gzopen(path, "rb")), buf_(1 << 20) {
        if (gz_) gzbuffer(gz_, 1 << 20);
    }
    ~LineReader() {
        if (gz_) gzclose(gz_);
    }
    bool ok() const { return gz_ != nullptr; }
    bool failed() const { return failed_; }
    /* next line without its terminator (\n or \r\n); false at end of file */
    bool next(std::string &line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos_ == len_) {
                const int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
                if (n < 0) {
                    failed_ = true;
                    return false;
                }
                if (n == 0) { /* end of file: a last line without terminator still counts */
                    if (any && !line.empty() && line.back() == '\r') line.pop_back();
                    return any;
                }
                pos_ = 0;
                len_ = (size_t)n;
            }
            const char *s = buf_.data() + pos_;
            const char *nl = (const char *)memchr(s, '\n', len_ - pos_);
            if (nl) {
                line.append(s, (size_t)(nl - s));
                pos_ += (size_t)(nl - s) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(s, len_ - pos_);
            pos_ = len_;
            any = true;
        }
    }

  private:
    gzFile gz_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool failed_ = false;
};

void split(const std::string &s, char sep, std::vector<std::string> &out) {
    out.clear();
    size_t b = 0;
    for (;;) {
        const size_t e = s.find(sep, b);
        if (e == std::string::npos) {
            out.emplace_back(s, b);
            return;
        }
        out.emplace_back(s, b, e - b);
        b = e + 1;
    }
}

void split_sv(std::string_view s, char sep, std::vector<std::string_view> &out) {
    out.clear();
    size_t b = 0;
    for (;;) {
        const size_t e = s.find(sep, b);
        if (e == std::string_view::npos) {
            out.push_back(s.substr(b));
            return;
        }
        out.push_back(s.substr(b, e - b));
        b = e + 1;
    }
}

bool parse_u64_sv(std::string_view s, uint64_t &v) {
    if (s.empty()) return false;
    const auto r = std::from_chars(s.data(), s.data() + s.size(), v);
    return r.ec == std::errc() && r.ptr == s.data() + s.size();
}

bool parse_u64(const std::string &s, uint64_t &v) {
    if (s.empty()) return false;
    const auto r = std::from_chars(s.data(), s.data() + s.size(), v);
    return r.ec == std::errc() && r.ptr == s.data() + s.size();
}

} // namespace

/* ------------------------------------------------------------------------------------------ genome */
/* a byte vector whose resize() leaves the new bytes uninitialised (the parallel FASTA loader fills them from several threads) */
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind {
        using other = NoInitAlloc<U>;
    };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
typedef std::vector<uint8_t, NoInitAlloc<uint8_t>> SeqBytes;

struct avf_genome {
    std::vector<std::string> names;
    std::vector<SeqBytes> seqs;
    std::unordered_map<std::string, uint32_t> index;
};

/* ------------------------------------------------------------------------------------------ feed */
struct avf_feed {
    uint32_t k = 2;              /* inputs: 2 for compare (truth, query), the number of VCFs for merge */
    bool is_merge = false;
    avk_region_batch batch;      /* compare feeds */
    avk_multi_batch multi;       /* merge feeds */
    /* resize() of these leaves new elements uninitialised: the chromosomes' parts are copied in by several threads */
    std::vector<uint64_t, NoInitAlloc<uint64_t>> region_id, start, end, in_off, t_off, q_off, var_pos, a0_off, a1_off, var_record;
    std::vector<uint32_t, NoInitAlloc<uint32_t>> contig_idx, in_cnt, t_cnt, q_cnt, var_raw, a0_len, a1_len, var_alt;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> var_type, var_zyg, alleles;
    std::vector<uint64_t> loaded;
};

namespace {

/* one parsed call: Variant + PhasedZygosity (variants.rs:73-91, phase_enums.rs) */
struct Call {
    uint64_t pos;      /* 0-based */
    std::string a0, a1;
    uint32_t raw_space;
    uint8_t type, zyg;
    uint64_t record;   /* index of the data line in its file */
    uint32_t alt_index;
};

/* the calls of one chromosome in file order, as the chunks the reader's threads produced them (no copy into one list);
 * base[k] is added to Call::record of the calls of chunk k */
struct CallList {
    std::vector<std::vector<Call>> chunks;
    std::vector<uint64_t> base;
    bool open = false; /* the last chunk takes single calls */
    void push(Call &&c) {
        if (!open) {
            chunks.emplace_back();
            base.push_back(0);
            open = true;
        }
        chunks.back().push_back(std::move(c));
    }
    void add_chunk(std::vector<Call> &&calls, uint64_t record_base) {
        if (calls.empty()) return;
        chunks.push_back(std::move(calls));
        base.push_back(record_base);
        open = false;
    }
};
typedef std::unordered_map<std::string, CallList> CallMap;

struct Interval1 { /* noodles Interval: 1-based, inclusive */
    uint64_t start, end;
};

struct LoadedBed {
    std::vector<std::string> chroms; /* insertion order (IndexMap) */
    std::vector<std::vector<Interval1>> intervals;
};

/* LoadedBed::preload_bed_file (noodles_helper.rs:48-86): chromosome order of first appearance, intervals sorted by
 * (start, end) when they are not already */
int load_bed(const char *path, LoadedBed &bed) {
    LineReader in(path);
    if (!in.ok()) return fail(AVK_E_ARG, "cannot open BED file %s", path);
    std::unordered_map<std::string, size_t> at;
    std::string line;
    std::vector<std::string> f;
    uint64_t lineno = 0;
    while (in.next(line)) {
        lineno += 1;
        if (line.empty()) continue;
        if (line[0] == '#' || line.compare(0, 5, "track") == 0 || line.compare(0, 7, "browser") == 0) continue;
        split(line, '\t', f);
        uint64_t s = 0, e = 0;
        if (f.size() < 3 || !parse_u64(f[1], s) || !parse_u64(f[2], e)) return fail(AVK_E_ARG, "%s:%llu: malformed BED record", path, (unsigned long long)lineno);
        auto it = at.find(f[0]);
        size_t k;
        if (it == at.end()) {
            k = bed.chroms.size();
            at.emplace(f[0], k);
            bed.chroms.push_back(f[0]);
            bed.intervals.emplace_back();
        } else k = it->second;
        /* BED [s, e) -> Interval [s+1, e] */
        bed.intervals[k].push_back(Interval1{s + 1, e});
    }
    if (in.failed()) return fail(AVK_E_ARG, "read error in %s", path);
    for (auto &iv : bed.intervals) {
        auto less = [](const Interval1 &a, const Interval1 &b) { return a.start != b.start ? a.start < b.start : a.end < b.end; };
        if (!std::is_sorted(iv.begin(), iv.end(), less)) std::sort(iv.begin(), iv.end(), less); /* sort_by_key: stable; equal keys are identical */
    }
    return 0;
}

/* get_variant_type (region_generation.rs:715-758) + the Variant constructors' length rules (variants.rs:104-383).
 * returns 0 and the type, 1 = skip this call (unsupported SV kinds), < 0 = error */
int variant_type_of(const std::string &svtype, bool has_svtype, bool has_trid, size_t l0, size_t l1, uint8_t &type) {
    if (has_svtype) {
        if (svtype == "BND" || svtype == "DUP") return 1; /* SvBreakend / SvDuplication: `continue` at :634-637 */
        if (svtype == "DEL") {
            if (l0 <= 1) return fail(-1, "SV deletion: reference must have length > 1");
            if (l1 > l0) return fail(-1, "SV deletion ALT length must be <= REF length");
            type = AVK_VT_SV_DELETION;
            return 0;
        }
        if (svtype == "INS") {
            if (l1 < l0) return fail(-1, "SV insertion ALT length must be >= REF length");
            if (l0 == 0) return fail(-1, "allele0 is empty (length = 0)");
            type = AVK_VT_SV_INSERTION;
            return 0;
        }
        return fail(-1, "Unsupported SVTYPE detected: %s", svtype.c_str());
    }
    if (has_trid) {
        if (l0 == 0 || l1 == 0) return fail(-1, "tandem repeat allele is empty (length = 0)");
        type = l1 < l0 ? AVK_VT_TR_CONTRACTION : AVK_VT_TR_EXPANSION;
        return 0;
    }
    if (l0 == 0 || l1 == 0) return fail(-1, "cannot have alleles with 0 length");
    if (l0 == 1 && l1 == 1) type = AVK_VT_SNV;
    else if (l0 == 1) type = AVK_VT_INSERTION;
    else if (l1 == 1) type = AVK_VT_DELETION;
    else type = AVK_VT_INDEL;
    return 0;
}

/* All calls of one sample on every chromosome: parse_variant + parse_genotype (region_generation.rs:563-712).
 * calls[chrom] keeps file order. */
/* per-thread scratch of the record parser: the fields are views into the line, nothing is copied until a call is made */
struct VcfScratch {
    std::vector<std::string> f; /* the #CHROM line */
    std::vector<std::string_view> v, alts, fmt, sv, info;
    std::string chrom;
};

/* the #CHROM line: which column holds the sample */
int vcf_sample_column(const char *path, const std::string &line, const char *sample, VcfScratch &w, long &sample_col) {
    split(line, '\t', w.f);
    if (w.f.size() < 10) return fail(AVK_E_ARG, "%s has no sample columns", path);
    sample_col = -1;
    if (!sample || !*sample) sample_col = 9; /* get_vcf_sample_name(.., 0) */
    else {
        for (size_t k = 9; k < w.f.size(); ++k)
            if (w.f[k] == sample) {
                sample_col = (long)k;
                break;
            }
        if (sample_col < 0) return fail(AVK_E_ARG, "Sample name \"%s\" was not found in %s", sample, path);
    }
    return 0;
}

/* One data line -> zero, one or two calls (load_variants_in_region / parse_variant / parse_genotype, region_generation.rs:489-758),
 * handed to emit(chrom, call).  rec = index of the line among the data lines (for messages and provenance). */
template <class Emit> int vcf_parse_record(const char *path, std::string_view line, long sample_col, bool enable_trimming, uint64_t rec, VcfScratch &w, Emit &&emit) {
    std::vector<std::string_view> &f = w.v, &alts = w.alts, &fmt = w.fmt, &sv = w.sv, &info = w.info;
    split_sv(line, '\t', f);
    if ((long)f.size() <= sample_col) return fail(AVK_E_ARG, "%s: record %llu has too few columns", path, (unsigned long long)rec);
    uint64_t pos1 = 0;
    if (!parse_u64_sv(f[1], pos1) || pos1 == 0) return fail(AVK_E_ARG, "%s: record %llu: Missing POS", path, (unsigned long long)rec);
    const std::string_view ref_seq = f[3];
    /* sample GT */
    split_sv(f[8], ':', fmt);
    long gt_at = -1;
    for (size_t k = 0; k < fmt.size(); ++k)
        if (fmt[k] == "GT") {
            gt_at = (long)k;
            break;
        }
    if (gt_at < 0)
        return fail(AVK_E_ARG, "%s: record %llu (%s:%llu): Missing GT", path, (unsigned long long)rec, std::string(f[0]).c_str(), (unsigned long long)pos1);
    split_sv(f[(size_t)sample_col], ':', sv);
    if ((long)sv.size() <= gt_at || sv[(size_t)gt_at] == "." || sv[(size_t)gt_at].empty()) return 0; /* GT = '.': a no-op (:583-586) */
    const std::string_view gt = sv[(size_t)gt_at];
    /* parse_genotype (:660-712) */
    uint64_t idx[2] = {0, 0};
    bool phased = false;
    {
        size_t n_alleles = 0, b = 0;
        for (size_t k = 0; k <= gt.size(); ++k) {
            if (k == gt.size() || gt[k] == '/' || gt[k] == '|') {
                if (k < gt.size() && gt[k] == '|') phased = true;
                if (n_alleles >= 2) return fail(AVK_E_ARG, "%s: record %llu: allele.len() != [1, 2]: %s", path, (unsigned long long)rec, std::string(gt).c_str());
                const std::string_view a = gt.substr(b, k - b);
                uint64_t v = 0;
                if (a != "." && !parse_u64_sv(a, v)) return fail(AVK_E_ARG, "%s: record %llu: malformed GT %s", path, (unsigned long long)rec, std::string(gt).c_str());
                idx[n_alleles++] = v; /* '.' is treated as a reference call */
                b = k + 1;
            }
        }
        if (n_alleles == 1) idx[1] = idx[0]; /* hemizygous is treated as homozygous */
    }
    std::pair<uint64_t, uint8_t> picks[2];
    int n_picks = 0;
    if (idx[0] == idx[1]) {
        if (idx[0] != 0) picks[n_picks++] = {idx[0], (uint8_t)AVK_ZYG_HOM_ALT};
    } else {
        const uint8_t ap1 = phased ? AVK_ZYG_PHASED_HET10 : AVK_ZYG_UNPHASED_HET, ap2 = phased ? AVK_ZYG_PHASED_HET01 : AVK_ZYG_UNPHASED_HET;
        if (idx[0] != 0) picks[n_picks++] = {idx[0], ap1};
        if (idx[1] != 0) picks[n_picks++] = {idx[1], ap2};
    }
    if (n_picks == 0) return 0;
    if (f[4] == "." || f[4].empty()) alts.clear();
    else split_sv(f[4], ',', alts);
    /* INFO: SVTYPE and TRID */
    bool has_svtype = false, has_trid = false;
    std::string svtype;
    if (f[7] != "." && !f[7].empty()) {
        split_sv(f[7], ';', info);
        for (const std::string_view kv : info) {
            if (kv.compare(0, 7, "SVTYPE=") == 0) {
                has_svtype = true;
                svtype.assign(kv.substr(7));
            } else if (kv.compare(0, 5, "TRID=") == 0 && kv.size() > 5) has_trid = true;
        }
    }
    for (int p = 0; p < n_picks; ++p) {
        const uint64_t alt_index = picks[p].first;
        if (alt_index > alts.size()) return fail(AVK_E_ARG, "%s: record %llu: GT refers to ALT %llu of %zu", path, (unsigned long long)rec, (unsigned long long)alt_index, alts.size());
        const std::string_view alt = alts[alt_index - 1];
        if (alt == "*") continue;            /* effectively a reference allele (:597-600) */
        if (!alt.empty() && alt[0] == '<') continue; /* symbolic: needs sequence-resolved (:604-607) */
        size_t rl = ref_seq.size(), al = alt.size();
        const size_t raw_space = std::max(rl, al); /* before trimming (:612) */
        while (enable_trimming && rl > 1 && al > 1 && ref_seq[rl - 1] == alt[al - 1]) {
            rl -= 1;
            al -= 1;
        }
        if (rl > 10000 || al > 10000) continue; /* allele_size_limit (:621-626) */
        uint8_t type = 0;
        const int rc = variant_type_of(svtype, has_svtype, has_trid, rl, al, type);
        if (rc == 1) continue;
        if (rc < 0)
            return fail(AVK_E_ARG, "%s: record %llu (%s:%llu): %s", path, (unsigned long long)rec, std::string(f[0]).c_str(), (unsigned long long)pos1, std::string(t_error).c_str());
        Call c;
        c.pos = pos1 - 1;
        c.a0.assign(ref_seq.data(), rl);
        c.a1.assign(alt.data(), al);
        c.raw_space = (uint32_t)raw_space;
        c.type = type;
        c.zyg = picks[p].second;
        c.record = rec;
        c.alt_index = (uint32_t)alt_index;
        if (w.chrom.size() != f[0].size() || memcmp(w.chrom.data(), f[0].data(), f[0].size()) != 0) w.chrom.assign(f[0]);
        emit(w.chrom, std::move(c));
    }
    return 0;
}

/* the whole file line by line on the calling thread: the reference behaviour, and the source of every error message */
int load_vcf_sequential(const char *path, const char *sample, bool enable_trimming, CallMap &calls) {
    LineReader in(path);
    if (!in.ok()) return fail(AVK_E_ARG, "Error while opening %s", path);
    std::string line;
    VcfScratch w;
    long sample_col = -1;
    uint64_t record = 0;
    bool have_header = false;
    while (in.next(line)) {
        if (line.empty()) continue;
        if (line[0] == '#') {
            if (line.compare(0, 6, "#CHROM") == 0) {
                const int rc = vcf_sample_column(path, line, sample, w, sample_col);
                if (rc) return rc;
                have_header = true;
            }
            continue;
        }
        if (!have_header) return fail(AVK_E_ARG, "%s: data line before the #CHROM header", path);
        const int rc = vcf_parse_record(path, line, sample_col, enable_trimming, record++, w, [&](const std::string &chrom, Call &&c) { calls[chrom].push(std::move(c)); });
        if (rc) return rc;
    }
    if (in.failed()) return fail(AVK_E_ARG, "read error in %s", path);
    if (!have_header) return fail(AVK_E_ARG, "%s has no #CHROM header line", path);
    return 0;
}

/* the block table of a BGZF file (gzip members with the BC extra subfield): offset of the deflate payload, its length, the
 * uncompressed size and checksum; false when the file is anything else */
struct BgzfBlk {
    size_t data, len;
    uint32_t isize, crc;
};
bool scan_bgzf_blocks(const uint8_t *d, size_t n, std::vector<BgzfBlk> &blks) {
    for (size_t at = 0; at < n;) {
        if (n - at < 18 || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8 || !(d[at + 3] & 4)) return false;
        const size_t xlen = d[at + 10] | ((size_t)d[at + 11] << 8);
        if (n - at < 12 + xlen + 8) return false;
        size_t bsize = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) { /* extra subfields: SI1 SI2 SLEN data */
            const size_t slen = d[x + 2] | ((size_t)d[x + 3] << 8);
            if (d[x] == 'B' && d[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = (d[x + 4] | ((size_t)d[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > n - at) return false;
        if (d[at + 3] & ~4) return false; /* name / comment / header crc fields: not what bgzip writes */
        BgzfBlk b;
        b.data = at + 12 + xlen;
        b.len = bsize - (12 + xlen) - 8;
        b.crc = d[at + bsize - 8] | ((uint32_t)d[at + bsize - 7] << 8) | ((uint32_t)d[at + bsize - 6] << 16) | ((uint32_t)d[at + bsize - 5] << 24);
        b.isize = d[at + bsize - 4] | ((uint32_t)d[at + bsize - 3] << 8) | ((uint32_t)d[at + bsize - 2] << 16) | ((uint32_t)d[at + bsize - 1] << 24);
        if (b.isize > 65536) return false;
        if (b.isize) blks.push_back(b); /* empty blocks (the end-of-file marker) carry nothing */
        at += bsize;
    }
    return true;
}

/* BGZF files (what bgzip / htslib write, and what the reference requires for its tabix queries): the blocks are independent gzip
 * members, so GROUPS of blocks are inflated AND parsed by the worker threads.  A group's text starts and ends in the middle of a line:
 * the workers parse the whole lines inside, the fragments at both ends are put together afterwards (one line per group boundary).
 * Returns 1 when the file is not BGZF or anything is irregular — the caller then reads it the ordinary way. */
int load_vcf_bgzf(const char *path, const char *sample, bool enable_trimming, CallMap &calls, size_t n_workers) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return 1;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28) {
        close(fd);
        return 1;
    }
    const size_t n = (size_t)st.st_size;
    const uint8_t *d = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (d == (const uint8_t *)MAP_FAILED) return 1;
    struct Unmap {
        const uint8_t *p;
        size_t n;
        ~Unmap() { munmap((void *)p, n); }
    } unmap{d, n};
    std::vector<BgzfBlk> blks;
    if (!scan_bgzf_blocks(d, n, blks)) return 1;
    typedef BgzfBlk Blk;
    auto inflate_block = [&](const Blk &b, z_stream &zs, std::string &out) -> bool {
        const size_t at = out.size();
        out.resize(at + b.isize);
        if (inflateReset(&zs) != Z_OK) return false;
        zs.next_in = (Bytef *)(d + b.data);
        zs.avail_in = (uInt)b.len;
        zs.next_out = (Bytef *)&out[at];
        zs.avail_out = b.isize;
        if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) return false;
        return (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)&out[at], b.isize) == b.crc;
    };
    auto crc_of = [](const void *p, size_t n) { /* (libdeflate's CRC-32 is several times zlib 1.2.11's) */
        return avf_bgzf::libdeflate().ok ? avf_bgzf::libdeflate().crc32(0, p, n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)p, (uInt)n);
    };
    auto inflate_block_w = [&](const Blk &b, avf_bgzf::Inflater &inf, std::string &out) -> bool { /* the workers': libdeflate when the system has it */
        const size_t at = out.size();
        out.resize(at + b.isize);
        return inf.run(d + b.data, b.len, &out[at], b.isize) && crc_of(&out[at], b.isize) == b.crc;
    };
    /* the header: inflate from the start until the #CHROM line is complete */
    long sample_col = -1;
    {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) return 1;
        std::string head;
        bool found = false, bad = false;
        size_t scanned = 0;
        VcfScratch w;
        for (size_t k = 0; k < blks.size() && !found && !bad; ++k) {
            if (!inflate_block(blks[k], zs, head)) {
                bad = true;
                break;
            }
            for (;;) {
                const size_t nl = head.find('\n', scanned);
                if (nl == std::string::npos) break;
                size_t len = nl - scanned;
                if (len && head[scanned + len - 1] == '\r') len -= 1;
                if (len) {
                    if (head[scanned] != '#') { /* a data line before #CHROM: the ordinary reader words the error */
                        bad = true;
                        break;
                    }
                    if (len >= 6 && head.compare(scanned, 6, "#CHROM") == 0) {
                        const int rc = vcf_sample_column(path, head.substr(scanned, len), sample, w, sample_col);
                        inflateEnd(&zs);
                        if (rc) return rc;
                        found = true;
                        break;
                    }
                }
                scanned = nl + 1;
            }
        }
        if (!found) {
            inflateEnd(&zs);
            return 1;
        }
        inflateEnd(&zs);
    }
    const auto t_begin = std::chrono::steady_clock::now();
    size_t group_blocks = 16;
    if (const char *e = getenv("AVF_VCF_GROUP")) group_blocks = std::max<size_t>(1, (size_t)strtoull(e, nullptr, 10)); /* tests: force group boundaries */
    const size_t n_groups = (blks.size() + group_blocks - 1) / group_blocks;
    struct GroupOut {
        std::vector<std::string> chroms;
        std::vector<std::vector<Call>> lists;
        uint64_t n_records = 0;
        std::string head, tail; /* the fragments before the first and after the last line feed; head = everything when there is none */
        bool has_newline = false, has_chrom = false, has_data = false, chrom_after_data = false;
    };
    std::vector<GroupOut> outs(n_groups);
    std::atomic<size_t> next{0};
    std::atomic<bool> irregular{false};
    auto emit_into = [](GroupOut &out, size_t &last, const std::string &chrom, Call &&c) {
        if (last == (size_t)-1 || out.chroms[last] != chrom) {
            last = (size_t)-1;
            for (size_t q = 0; q < out.chroms.size(); ++q)
                if (out.chroms[q] == chrom) last = q;
            if (last == (size_t)-1) {
                out.chroms.push_back(chrom);
                out.lists.emplace_back();
                last = out.chroms.size() - 1;
            }
        }
        out.lists[last].push_back(std::move(c));
    };
    auto worker = [&] {
        avf_bgzf::Inflater zs;
        if (!zs.ok()) {
            irregular.store(true);
            return;
        }
        VcfScratch w;
        std::string text, line;
        for (size_t g = next.fetch_add(1); g < n_groups && !irregular.load(std::memory_order_relaxed); g = next.fetch_add(1)) {
            GroupOut &out = outs[g];
            text.clear();
            bool ok = true;
            for (size_t k = g * group_blocks; k < std::min(blks.size(), (g + 1) * group_blocks) && ok; ++k) ok = inflate_block_w(blks[k], zs, text);
            if (!ok) {
                irregular.store(true);
                break;
            }
            const size_t first_nl = text.find('\n');
            if (first_nl == std::string::npos) {
                out.head = text;
                continue;
            }
            out.has_newline = true;
            const size_t last_nl = text.rfind('\n');
            out.head.assign(text, 0, first_nl);
            out.tail.assign(text, last_nl + 1, std::string::npos);
            size_t last = (size_t)-1;
            const char *p = text.data() + first_nl + 1, *end = text.data() + last_nl + 1;
            while (p < end) {
                const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                size_t len = (size_t)(nl - p);
                if (len && p[len - 1] == '\r') len -= 1;
                if (len) {
                    if (p[0] == '#') {
                        if (len >= 6 && memcmp(p, "#CHROM", 6) == 0) {
                            out.has_chrom = true;
                            if (out.has_data) out.chrom_after_data = true;
                        }
                    } else {
                        out.has_data = true;
                        const int rc = vcf_parse_record(path, std::string_view(p, len), sample_col, enable_trimming, out.n_records, w,
                                                        [&](const std::string &chrom, Call &&c) { emit_into(out, last, chrom, std::move(c)); });
                        out.n_records += 1;
                        if (rc) {
                            irregular.store(true);
                            break;
                        }
                    }
                }
                p = nl + 1;
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < std::min(n_workers, std::max<size_t>(n_groups, 1)); ++t) pool.emplace_back(worker);
        worker();
        for (std::thread &t : pool) t.join();
    }
    if (irregular.load()) return 1;
    const auto t_parsed = std::chrono::steady_clock::now();
    /* in file order: the line put together at a group's front, then the group's own lines; record indices become file-wide */
    uint64_t base = 0;
    bool data_seen = false;
    std::string carry;
    VcfScratch w;
    GroupOut joint;
    auto boundary_line = [&](std::string &text) -> int { /* one complete line made of fragments */
        if (!text.empty() && text.back() == '\r') text.pop_back();
        if (text.empty()) return 0;
        if (text[0] == '#') {
            if (text.compare(0, 6, "#CHROM") == 0 && data_seen) return 1;
            return 0;
        }
        data_seen = true;
        size_t last = (size_t)-1;
        const int rc = vcf_parse_record(path, text, sample_col, enable_trimming, base, w, [&](const std::string &chrom, Call &&c) { emit_into(joint, last, chrom, std::move(c)); });
        base += 1;
        if (rc) return 1;
        for (size_t q = 0; q < joint.chroms.size(); ++q)
            for (Call &c : joint.lists[q]) calls[joint.chroms[q]].push(std::move(c));
        joint.chroms.clear();
        joint.lists.clear();
        return 0;
    };
    for (size_t g = 0; g < n_groups; ++g) {
        GroupOut &o = outs[g];
        carry += o.head;
        if (!o.has_newline) continue;
        if (boundary_line(carry)) {
            calls.clear();
            return 1;
        }
        carry = o.tail;
        if ((o.has_chrom && data_seen) || o.chrom_after_data) {
            calls.clear();
            return 1;
        }
        data_seen = data_seen || o.has_data;
        for (size_t q = 0; q < o.chroms.size(); ++q) calls[o.chroms[q]].add_chunk(std::move(o.lists[q]), base); /* the list is handed over as it is */
        base += o.n_records;
        GroupOut().chroms.swap(o.chroms);
        std::vector<std::vector<Call>>().swap(o.lists);
    }
    if (boundary_line(carry)) { /* a last line without a line feed */
        calls.clear();
        return 1;
    }
    if (getenv("AVF_TIMING"))
        fprintf(stderr, "[avf] vcf %s: %zu BGZF blocks in %zu groups, %llu records; inflate + parse %.3f s, put together %.3f s\n", path, blks.size(), n_groups,
                (unsigned long long)base, std::chrono::duration<double>(t_parsed - t_begin).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_parsed).count());
    return 0;
}

/* The file is decompressed on the calling thread and cut into blocks of whole lines; worker threads parse the blocks, and the calls
 * are put together in file order.  Anything irregular (an error, a #CHROM line after the first data line) is left to the sequential
 * reader above, which then reports it the usual way. */
int load_vcf(const char *path, const char *sample, bool enable_trimming, CallMap &calls) {
    const unsigned hw = avk_usable_cpus();
    const size_t n_workers = std::min<size_t>(hw > 1 ? hw - 1 : 0, 8);
    if (n_workers < 2 || getenv("AVF_SEQUENTIAL_VCF")) return load_vcf_sequential(path, sample, enable_trimming, calls);
    if (!getenv("AVF_NO_BGZF_GROUPS")) {
        const int rc = load_vcf_bgzf(path, sample, enable_trimming, calls, std::min<size_t>(hw, 16));
        if (rc != 1) return rc; /* done, or a header error worded the usual way */
        calls.clear();
    }
    gzFile gz = gzopen(path, "rb");
    if (!gz) return fail(AVK_E_ARG, "Error while opening %s", path);
    gzbuffer(gz, 1 << 20);
    size_t block_bytes = 4u << 20;
    if (const char *e = getenv("AVF_VCF_BLOCK")) block_bytes = std::max<size_t>(16, (size_t)strtoull(e, nullptr, 10)); /* tests: force block boundaries */

    struct BlockOut {
        std::vector<std::string> chroms;
        std::vector<std::vector<Call>> lists;
        uint64_t n_records = 0;
    };
    std::vector<std::unique_ptr<std::string>> blocks; /* block k's text, released once parsed */
    std::vector<std::unique_ptr<BlockOut>> outs;
    std::mutex mu;
    std::condition_variable cv_work, cv_room;
    size_t next_block = 0, in_flight = 0;
    bool done_reading = false;
    std::atomic<bool> irregular{false};
    long sample_col = -1;

    auto worker = [&] {
        VcfScratch w;
        std::string line;
        for (;;) {
            size_t k;
            std::string *text;
            BlockOut *out;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_work.wait(lock, [&] { return next_block < blocks.size() || done_reading; });
                if (next_block >= blocks.size()) return;
                k = next_block++;
                text = blocks[k].get();
                out = outs[k].get();
            }
            if (!irregular.load(std::memory_order_relaxed)) {
                size_t last = (size_t)-1;
                const char *p = text->data(), *end = p + text->size();
                while (p < end) {
                    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                    const char *le = nl ? nl : end;
                    size_t len = (size_t)(le - p);
                    if (len && p[len - 1] == '\r') len -= 1;
                    if (len) {
                        if (p[0] == '#') {
                            if (len >= 6 && memcmp(p, "#CHROM", 6) == 0) irregular.store(true);
                        } else {
                            const int rc = vcf_parse_record(path, std::string_view(p, len), sample_col, enable_trimming, out->n_records, w, [&](const std::string &chrom, Call &&c) {
                                if (last == (size_t)-1 || out->chroms[last] != chrom) {
                                    last = (size_t)-1;
                                    for (size_t q = 0; q < out->chroms.size(); ++q)
                                        if (out->chroms[q] == chrom) last = q;
                                    if (last == (size_t)-1) {
                                        out->chroms.push_back(chrom);
                                        out->lists.emplace_back();
                                        last = out->chroms.size() - 1;
                                    }
                                }
                                out->lists[last].push_back(std::move(c));
                            });
                            out->n_records += 1;
                            if (rc) {
                                irregular.store(true);
                                break;
                            }
                        }
                    }
                    p = nl ? nl + 1 : end;
                }
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                blocks[k].reset(); /* the text is no longer needed */
                in_flight -= 1;
            }
            cv_room.notify_one();
        }
    };
    std::vector<std::thread> pool;

    /* the reader: header lines first (they fix the sample column), then blocks of whole lines */
    std::string pending; /* bytes read but not yet handed out */
    std::vector<char> buf(1 << 20);
    bool have_header = false, in_header = true, read_failed = false;
    VcfScratch hw_scratch;
    int rc_header = 0;
    auto hand_out = [&](std::string &&text) {
        std::unique_lock<std::mutex> lock(mu);
        cv_room.wait(lock, [&] { return in_flight < 4 * n_workers; }); /* bounds the decompressed text held in memory */
        blocks.emplace_back(new std::string(std::move(text)));
        outs.emplace_back(new BlockOut());
        in_flight += 1;
        lock.unlock();
        cv_work.notify_one();
    };
    for (;;) {
        const int n = gzread(gz, buf.data(), (unsigned)buf.size());
        if (n < 0) {
            read_failed = true;
            break;
        }
        if (n == 0) break;
        pending.append(buf.data(), (size_t)n);
        if (in_header) { /* consume whole header / empty lines from the front */
            size_t at = 0;
            while (in_header) {
                const size_t nl = pending.find('\n', at);
                if (nl == std::string::npos) break;
                size_t len = nl - at;
                if (len && pending[at + len - 1] == '\r') len -= 1;
                if (len == 0 || pending[at] == '#') {
                    if (len >= 6 && pending.compare(at, 6, "#CHROM") == 0) {
                        rc_header = vcf_sample_column(path, pending.substr(at, len), sample, hw_scratch, sample_col);
                        if (rc_header) break;
                        have_header = true;
                    }
                    at = nl + 1;
                } else in_header = false; /* first data line */
            }
            pending.erase(0, at);
            if (rc_header) break;
            if (!in_header) {
                if (!have_header) break; /* data before #CHROM: the sequential reader words the error */
                for (size_t t = 0; t < n_workers; ++t) pool.emplace_back(worker);
            }
        }
        if (!in_header && pending.size() >= block_bytes) {
            const size_t cut = pending.rfind('\n');
            if (cut != std::string::npos) {
                std::string text(pending, 0, cut + 1);
                pending.erase(0, cut + 1);
                hand_out(std::move(text));
            }
        }
        if (irregular.load(std::memory_order_relaxed)) break;
    }
    gzclose(gz);
    if (rc_header) return rc_header;
    if (!in_header && have_header && !pending.empty() && !read_failed) hand_out(std::move(pending));
    {
        std::lock_guard<std::mutex> lock(mu);
        done_reading = true;
    }
    cv_work.notify_all();
    for (std::thread &t : pool) t.join();
    if (read_failed || irregular.load() || in_header || !have_header) {
        calls.clear();
        return load_vcf_sequential(path, sample, enable_trimming, calls);
    }
    /* in file order: block after block; record indices become file-wide */
    uint64_t base = 0;
    for (std::unique_ptr<BlockOut> &o : outs) {
        for (size_t q = 0; q < o->chroms.size(); ++q) calls[o->chroms[q]].add_chunk(std::move(o->lists[q]), base);
        base += o->n_records;
        o.reset();
    }
    return 0;
}

/* ryu's shortest round-trip text of an f64, as the csv crate writes it */
std::string fmt_f64(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
    std::string s(buf, r.ptr);
    std::string out;
    size_t i = 0;
    if (s[0] == '-') {
        out.push_back('-');
        i = 1;
    }
    const size_t epos = s.find('e');
    std::string digits;
    for (size_t k = i; k < epos; ++k)
        if (s[k] != '.') digits.push_back(s[k]);
    const int exp10 = std::stoi(s.substr(epos + 1));
    const int len = (int)digits.size();
    const int k = exp10 - (len - 1); /* value = digits * 10^k */
    const int kk = len + k;
    if (0 <= k && kk <= 16) {
        out += digits;
        out.append((size_t)k, '0');
        out += ".0";
    } else if (0 < kk && kk <= 16) {
        out.append(digits, 0, (size_t)kk);
        out.push_back('.');
        out.append(digits, (size_t)kk, std::string::npos);
    } else if (-5 < kk && kk <= 0) {
        out += "0.";
        out.append((size_t)(-kk), '0');
        out += digits;
    } else {
        out.push_back(digits[0]);
        if (len > 1) {
            out.push_back('.');
            out.append(digits, 1, std::string::npos);
        }
        out.push_back('e');
        out += std::to_string(kk - 1);
    }
    return out;
}

std::string csv_field(const std::string &s, char delim) {
    bool quote = false;
    for (char c : s)
        if (c == delim || c == '"' || c == '\n' || c == '\r') quote = true;
    if (!quote) return s;
    std::string out = "\"";
    for (char c : s) {
        if (c == '"') out.push_back('"');
        out.push_back(c);
    }
    out.push_back('"');
    return out;
}

} // namespace

std::string avf_fmt_f64_(double v) { return fmt_f64(v); } /* for the other translation units */
std::string avf_csv_field_(const std::string &s, char delim) { return csv_field(s, delim); }

extern "C" {

const char *avf_last_error(void) { return t_error.c_str(); }

/* Uncompressed FASTA files are mapped and parsed by several threads: header lines are found first ('>' at a line start), then
 * every sequence body is cut into pieces whose kept bytes (everything but line terminators) are counted and copied side by side.
 * Same result as the line-by-line reader below, which still serves gzip / BGZF files. */
static int genome_load_mapped(const char *fasta_path, const uint8_t *d, size_t n, avf_genome *g, bool upper) {
    const unsigned hw = avk_usable_cpus();
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, 32));
    auto parallel = [&](size_t items, const std::function<void(size_t)> &fn) {
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i = next.fetch_add(1); i < items; i = next.fetch_add(1)) fn(i);
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < std::min(n_threads, items); ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    };
    const bool timing = getenv("AVF_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[avf] fasta %s: %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    /* 1. header positions */
    size_t piece = 8u << 20;
    if (const char *e = getenv("AVF_FASTA_PIECE")) piece = std::max<size_t>(1, (size_t)strtoull(e, nullptr, 10)); /* tests: force piece boundaries */
    const size_t n_pieces = (n + piece - 1) / piece;
    std::vector<std::vector<size_t>> found(n_pieces);
    parallel(n_pieces, [&](size_t k) {
        const size_t lo = k * piece, hi = std::min(n, lo + piece);
        for (size_t at = lo; at < hi;) {
            const uint8_t *q = (const uint8_t *)memchr(d + at, '>', hi - at);
            if (!q) break;
            const size_t pos = (size_t)(q - d);
            if (pos == 0 || d[pos - 1] == '\n') found[k].push_back(pos);
            at = pos + 1;
        }
    });
    lap("header scan");
    std::vector<size_t> headers;
    for (const auto &f : found) headers.insert(headers.end(), f.begin(), f.end());
    /* anything but empty lines before the first header is an error */
    const size_t first = headers.empty() ? n : headers[0];
    for (size_t at = 0; at < first; ++at)
        if (d[at] != '\n' && !(d[at] == '\r' && (at + 1 == n || d[at + 1] == '\n'))) return fail(AVK_E_ARG, "%s: sequence before the first header", fasta_path);
    /* 2. names and bodies */
    struct Piece {
        size_t contig, lo, hi, kept, out;
    };
    std::vector<Piece> pieces;
    g->names.reserve(headers.size());
    for (size_t i = 0; i < headers.size(); ++i) {
        const size_t h = headers[i], end = i + 1 < headers.size() ? headers[i + 1] : n;
        const uint8_t *nl = (const uint8_t *)memchr(d + h, '\n', end - h);
        size_t eol = nl ? (size_t)(nl - d) : end, line_end = eol;
        if (line_end > h && d[line_end - 1] == '\r') line_end -= 1;
        size_t e = h + 1;
        while (e < line_end && d[e] != ' ' && d[e] != '\t') ++e;
        g->names.emplace_back((const char *)d + h + 1, e - h - 1);
        const size_t body = nl ? eol + 1 : end;
        for (size_t lo = body; lo < end; lo += piece) pieces.push_back(Piece{i, lo, std::min(end, lo + piece), 0, 0});
    }
    g->seqs.resize(headers.size());
    /* a byte is dropped when it is a line feed, or a carriage return right before a line feed or the end of the file */
    auto dropped = [&](size_t at) { return d[at] == '\n' || (d[at] == '\r' && (at + 1 == n || d[at + 1] == '\n')); };
    parallel(pieces.size(), [&](size_t k) {
        Piece &pc = pieces[k];
        size_t drop = 0;
        for (size_t at = pc.lo; at < pc.hi;) {
            const uint8_t *q = (const uint8_t *)memchr(d + at, '\n', pc.hi - at);
            if (!q) break;
            const size_t pos = (size_t)(q - d);
            drop += 1 + (pos > pc.lo && d[pos - 1] == '\r' ? 1 : 0);
            at = pos + 1;
        }
        if (d[pc.hi - 1] == '\r' && dropped(pc.hi - 1)) drop += 1; /* its line feed is the first byte of the next piece, or the file ends here */
        pc.kept = (pc.hi - pc.lo) - drop;
    });
    lap("count");
    std::vector<size_t> total(headers.size(), 0);
    for (Piece &pc : pieces) {
        pc.out = total[pc.contig];
        total[pc.contig] += pc.kept;
    }
    for (size_t i = 0; i < headers.size(); ++i) {
        g->seqs[i].resize(total[i]);
        /* 3 GB written once by many threads: with 4 KB pages that is 760,000 first-touch faults; huge pages where the system grants them on request
         * (transparent_hugepage = madvise) */
        const uintptr_t huge = (uintptr_t)2 << 20, lo = ((uintptr_t)g->seqs[i].data() + huge - 1) & ~(huge - 1), hi = ((uintptr_t)g->seqs[i].data() + total[i]) & ~(huge - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }
    lap("allocate");
    parallel(pieces.size(), [&](size_t k) {
        const Piece &pc = pieces[k];
        uint8_t *out = g->seqs[pc.contig].data() + pc.out;
        size_t at = pc.lo;
        while (at < pc.hi) {
            const uint8_t *q = (const uint8_t *)memchr(d + at, '\n', pc.hi - at);
            size_t stop = q ? (size_t)(q - d) : pc.hi; /* end of this line's bytes inside the piece */
            size_t keep_to = stop;
            if (keep_to > at && d[keep_to - 1] == '\r' && dropped(keep_to - 1)) keep_to -= 1;
            memcpy(out, d + at, keep_to - at);
            if (upper) /* soft-masked (lower-case) bases become upper case: see avf_genome_load_case */
                for (size_t j = 0; j < keep_to - at; ++j) out[j] = (uint8_t)(out[j] >= 'a' && out[j] <= 'z' ? out[j] - 32 : out[j]);
            out += keep_to - at;
            at = q ? stop + 1 : pc.hi;
        }
    });
    lap("copy");
    return 0;
}

int avf_genome_load(const char *fasta_path, avf_genome **out) { return avf_genome_load_case(fasta_path, 1, out); }

int avf_genome_load_case(const char *fasta_path, int upper_case, avf_genome **out) {
    if (!fasta_path || !out) return fail(AVK_E_ARG, "null argument");
    const bool upper = upper_case != 0;
    *out = nullptr;
    avf_genome *g = new avf_genome();
    /* plain text (no gzip magic): map it */
    {
        const int fd = open(fasta_path, O_RDONLY);
        if (fd < 0) {
            delete g;
            return fail(AVK_E_ARG, "cannot open FASTA file %s", fasta_path);
        }
        struct stat st;
        uint8_t magic[2] = {0, 0};
        const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
        const bool gz = regular && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (regular && gz) { /* bgzip-compressed (what faidx wants): the blocks are inflated by several threads, then parsed like a plain file */
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                const uint8_t *d = (const uint8_t *)m;
                std::vector<BgzfBlk> blks;
                SeqBytes raw;
                bool ok = scan_bgzf_blocks(d, (size_t)st.st_size, blks) && !blks.empty();
                if (ok) {
                    std::vector<size_t> at(blks.size() + 1, 0);
                    for (size_t k = 0; k < blks.size(); ++k) at[k + 1] = at[k] + blks[k].isize;
                    raw.resize(at.back());
                    const unsigned hw = avk_usable_cpus();
                    const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)32, blks.size()}));
                    std::atomic<size_t> next{0};
                    std::atomic<bool> bad{false};
                    auto work = [&] {
                        avf_bgzf::Inflater zs; /* libdeflate when the system has it */
                        if (!zs.ok()) {
                            bad.store(true);
                            return;
                        }
                        const bool ld = avf_bgzf::libdeflate().ok;
                        for (size_t k0 = next.fetch_add(64); k0 < blks.size() && !bad.load(std::memory_order_relaxed); k0 = next.fetch_add(64))
                            for (size_t k = k0; k < std::min(blks.size(), k0 + 64); ++k) {
                                const BgzfBlk &b = blks[k];
                                if (!zs.run(d + b.data, b.len, raw.data() + at[k], b.isize) ||
                                    (ld ? avf_bgzf::libdeflate().crc32(0, raw.data() + at[k], b.isize) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(raw.data() + at[k]), b.isize)) != b.crc)
                                    bad.store(true);
                            }
                    };
                    std::vector<std::thread> pool;
                    for (size_t t = 1; t < nt; ++t) pool.emplace_back(work);
                    work();
                    for (std::thread &t : pool) t.join();
                    ok = !bad.load();
                }
                munmap(m, (size_t)st.st_size);
                if (ok) {
                    close(fd);
                    const int rc = genome_load_mapped(fasta_path, raw.data(), raw.size(), g, upper);
                    if (rc) {
                        delete g;
                        return rc;
                    }
                    for (uint32_t i = 0; i < g->names.size(); ++i) g->index.emplace(g->names[i], i);
                    *out = g;
                    return 0;
                }
                /* a plain gzip stream, or a damaged file: the line reader below reads it (and reports it) */
            }
        }
        if (regular && !gz) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                const int rc = genome_load_mapped(fasta_path, (const uint8_t *)m, (size_t)st.st_size, g, upper);
                munmap(m, (size_t)st.st_size);
                close(fd);
                if (rc) {
                    delete g;
                    return rc;
                }
                for (uint32_t i = 0; i < g->names.size(); ++i) g->index.emplace(g->names[i], i);
                *out = g;
                return 0;
            }
        }
        close(fd);
    }
    LineReader in(fasta_path);
    if (!in.ok()) {
        delete g;
        return fail(AVK_E_ARG, "cannot open FASTA file %s", fasta_path);
    }
    std::string line;
    while (in.next(line)) {
        if (line.empty()) continue;
        if (line[0] == '>') {
            size_t e = 1;
            while (e < line.size() && line[e] != ' ' && line[e] != '\t') ++e;
            g->names.emplace_back(line, 1, e - 1);
            g->seqs.emplace_back();
            continue;
        }
        if (g->seqs.empty()) {
            delete g;
            return fail(AVK_E_ARG, "%s: sequence before the first header", fasta_path);
        }
        if (upper)
            for (char &ch : line) ch = (char)(ch >= 'a' && ch <= 'z' ? ch - 32 : ch);
        g->seqs.back().insert(g->seqs.back().end(), line.begin(), line.end());
    }
    if (in.failed()) {
        delete g;
        return fail(AVK_E_ARG, "read error in %s", fasta_path);
    }
    for (uint32_t i = 0; i < g->names.size(); ++i) g->index.emplace(g->names[i], i);
    *out = g;
    return 0;
}
uint32_t avf_genome_n_contigs(const avf_genome *g) { return g ? (uint32_t)g->names.size() : 0; }
const char *avf_genome_name(const avf_genome *g, uint32_t i) { return g && i < g->names.size() ? g->names[i].c_str() : nullptr; }
const uint8_t *avf_genome_seq(const avf_genome *g, uint32_t i) { return g && i < g->seqs.size() ? g->seqs[i].data() : nullptr; }
uint64_t avf_genome_len(const avf_genome *g, uint32_t i) { return g && i < g->seqs.size() ? g->seqs[i].size() : 0; }
void avf_genome_free(avf_genome *g) { delete g; }

/* RegionIterator::next (region_generation.rs:281-478) over k inputs; input i's variants of region m end up at
 * [in_off[m*k + i], +in_cnt[m*k + i]).  Both the compare and the merge iterator are this loop. */
struct avf_calls {
    CallMap by_chrom;
};

/* the VCFs are read side by side, like the reference's per-file parallel load (:324-347) */
static int load_call_sets(uint32_t k, const char *const *vcfs, const char *const *samples, int enable_trimming, std::vector<CallMap> &calls) {
    calls.assign(k, CallMap());
    std::vector<int> rcs(k, 0);
    std::vector<std::string> errs(k);
    {
        const uint32_t n_threads = std::min<uint32_t>(k, std::max(1u, std::min(8u, avk_usable_cpus())));
        std::atomic<uint32_t> next{0};
        auto work = [&] {
            for (uint32_t i = next.fetch_add(1); i < k; i = next.fetch_add(1)) {
                rcs[i] = load_vcf(vcfs[i], samples ? samples[i] : nullptr, enable_trimming != 0, calls[i]);
                if (rcs[i]) errs[i] = t_error;
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    }
    for (uint32_t i = 0; i < k; ++i)
        if (rcs[i]) {
            t_error = errs[i];
            return rcs[i];
        }
    return 0;
}

static int build_regions(uint32_t k, const CallMap *const *calls, const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap, avf_feed *f) {
    if (!regions_bed || !*regions_bed) return fail(AVK_E_ARG, "High confidence regions are currently required.");
    if (min_variant_gap == 0) return fail(AVK_E_ARG, "--min-variant-gap must be >0");
    LoadedBed bed;
    int rc = load_bed(regions_bed, bed);
    if (rc) return rc;

    f->k = k;
    f->loaded.assign(k, 0);
    struct Joint {
        uint32_t input;
        const Call *c;
        uint64_t record; /* index of the call's line among the data lines of its file */
    };
    const size_t n_chroms = bed.chroms.size();
    std::vector<uint32_t> contig_of(n_chroms);
    for (size_t ci = 0; ci < n_chroms; ++ci) {
        const auto git = g->index.find(bed.chroms[ci]);
        if (git == g->index.end()) return fail(AVK_E_ARG, "Chromosome %s was not found in reference genome", bed.chroms[ci].c_str());
        contig_of[ci] = git->second;
    }
    /* every chromosome on its own (a part has the feed's arrays, offsets relative to the part), several at a time */
    std::vector<std::unique_ptr<avf_feed>> parts(n_chroms);
    auto walk = [&](size_t ci) {
        avf_feed *part = parts[ci].get();
        part->loaded.assign(k, 0);
        const std::string &chrom = bed.chroms[ci];
        const std::vector<Interval1> &intervals = bed.intervals[ci];
        const uint32_t contig = contig_of[ci];
        const uint64_t chrom_length = g->seqs[contig].size();
        /* the span the reference queries through tabix: first interval's start to the LAST interval's end (:289-296) */
        const uint64_t zb_start = intervals.front().start - 1, zb_end = intervals.back().end;
        std::vector<Joint> joint;
        std::vector<std::vector<Joint>> vars(k);
        for (uint32_t input = 0; input < k; ++input) {
            const auto it = calls[input]->find(chrom);
            if (it == calls[input]->end()) continue;
            for (size_t ch = 0; ch < it->second.chunks.size(); ++ch)
                for (const Call &c : it->second.chunks[ch]) {
                    /* is_variant_contained (:764-778): first and last reference base inside the span */
                    const uint64_t last = c.pos + c.a0.size() - 1;
                    if (c.pos >= zb_start && c.pos < zb_end && last >= zb_start && last < zb_end) {
                        joint.push_back(Joint{input, &c, c.record + it->second.base[ch]});
                        part->loaded[input] += 1;
                    }
                }
        }
        std::stable_sort(joint.begin(), joint.end(), [](const Joint &a, const Joint &b) { return a.c->pos < b.c->pos; }); /* sort_by_key(position) */
        size_t head = 0; /* the deque's front */
        auto flush = [&](uint64_t ws, uint64_t we) {
            part->contig_idx.push_back(contig);
            part->start.push_back(ws);
            part->end.push_back(we);
            for (uint32_t input = 0; input < k; ++input) {
                part->in_off.push_back(part->var_pos.size());
                part->in_cnt.push_back((uint32_t)vars[input].size());
                for (const Joint &jv : vars[input]) {
                    const Call *c = jv.c;
                    part->var_pos.push_back(c->pos);
                    part->var_type.push_back(c->type);
                    part->var_zyg.push_back(c->zyg);
                    part->var_raw.push_back(c->raw_space);
                    part->a0_off.push_back(part->alleles.size());
                    part->a0_len.push_back((uint32_t)c->a0.size());
                    part->alleles.insert(part->alleles.end(), c->a0.begin(), c->a0.end());
                    part->a1_off.push_back(part->alleles.size());
                    part->a1_len.push_back((uint32_t)c->a1.size());
                    part->alleles.insert(part->alleles.end(), c->a1.begin(), c->a1.end());
                    part->var_record.push_back(jv.record);
                    part->var_alt.push_back(c->alt_index);
                }
                vars[input].clear();
            }
        };
        for (const Interval1 &iv : intervals) { /* the interval loop of RegionIterator::next (:373-470) */
            const uint64_t ib = iv.start - 1, ie = iv.end;
            bool have_window = false;
            uint64_t window_start = 0, window_end = 0;
            bool have_end = false; /* window_end survives a flush (":412 // window_end = None") */
            while (head < joint.size()) {
                const Joint &j = joint[head];
                const uint64_t vs = j.c->pos, ve = vs + j.c->a0.size();
                if (vs < ib) { /* Before: dropped */
                    head += 1;
                    continue;
                }
                if (vs >= ie) break; /* After: stays at the front for the next interval */
                head += 1;
                if (ve > ie) continue; /* Overlapping: dropped */
                /* Contained */
                if (have_end && vs >= window_end) { /* too far away: close the current block */
                    flush(window_start, window_end);
                    have_window = false;
                }
                if (!have_window) {
                    window_start = vs > min_variant_gap ? vs - min_variant_gap : 0;
                    have_window = true;
                }
                const uint64_t var_flank_end = std::min(vs + j.c->a0.size() + min_variant_gap, chrom_length);
                window_end = have_end ? std::max(window_end, var_flank_end) : var_flank_end;
                have_end = true;
                vars[j.input].push_back(j);
            }
            if (have_window && have_end) flush(window_start, window_end);
        }
    };
    const unsigned hw = avk_usable_cpus();
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)16, n_chroms}));
    auto parallel = [&](const std::function<void(size_t)> &fn) {
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t ci = next.fetch_add(1); ci < n_chroms; ci = next.fetch_add(1)) fn(ci);
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    };
    for (size_t ci = 0; ci < n_chroms; ++ci) parts[ci].reset(new avf_feed());
    parallel(walk);
    /* region ids run over the chromosomes in BED order (next_region_id); offsets become batch-wide */
    std::vector<uint64_t> r0(n_chroms + 1, 0), v0(n_chroms + 1, 0), a0(n_chroms + 1, 0);
    for (size_t ci = 0; ci < n_chroms; ++ci) {
        r0[ci + 1] = r0[ci] + parts[ci]->start.size();
        v0[ci + 1] = v0[ci] + parts[ci]->var_pos.size();
        a0[ci + 1] = a0[ci] + parts[ci]->alleles.size();
        for (uint32_t input = 0; input < k; ++input) f->loaded[input] += parts[ci]->loaded[input];
    }
    const uint64_t nr = r0[n_chroms], nv = v0[n_chroms], na = a0[n_chroms];
    f->region_id.resize(nr);
    f->contig_idx.resize(nr);
    f->start.resize(nr);
    f->end.resize(nr);
    f->in_off.resize(nr * k);
    f->in_cnt.resize(nr * k);
    f->var_pos.resize(nv);
    f->var_type.resize(nv);
    f->var_zyg.resize(nv);
    f->var_raw.resize(nv);
    f->a0_off.resize(nv);
    f->a0_len.resize(nv);
    f->a1_off.resize(nv);
    f->a1_len.resize(nv);
    f->var_record.resize(nv);
    f->var_alt.resize(nv);
    f->alleles.resize(na);
    parallel([&](size_t ci) {
        avf_feed *part = parts[ci].get();
        const uint64_t rb = r0[ci], vb = v0[ci], ab = a0[ci], n = part->start.size(), m = part->var_pos.size();
        auto copy = [](auto &dst, uint64_t at, const auto &src) {
            if (!src.empty()) memcpy(dst.data() + at, src.data(), src.size() * sizeof(src[0]));
        };
        for (uint64_t r = 0; r < n; ++r) f->region_id[rb + r] = rb + r;
        copy(f->contig_idx, rb, part->contig_idx);
        copy(f->start, rb, part->start);
        copy(f->end, rb, part->end);
        for (uint64_t q = 0; q < n * k; ++q) f->in_off[rb * k + q] = part->in_off[q] + vb;
        copy(f->in_cnt, rb * k, part->in_cnt);
        copy(f->var_pos, vb, part->var_pos);
        copy(f->var_type, vb, part->var_type);
        copy(f->var_zyg, vb, part->var_zyg);
        copy(f->var_raw, vb, part->var_raw);
        for (uint64_t q = 0; q < m; ++q) {
            f->a0_off[vb + q] = part->a0_off[q] + ab;
            f->a1_off[vb + q] = part->a1_off[q] + ab;
        }
        copy(f->a0_len, vb, part->a0_len);
        copy(f->a1_len, vb, part->a1_len);
        copy(f->var_record, vb, part->var_record);
        copy(f->var_alt, vb, part->var_alt);
        copy(f->alleles, ab, part->alleles);
        parts[ci].reset();
    });
    if (f->alleles.empty()) f->alleles.push_back(0);
    return 0;
}

static void fill_compare_batch(avf_feed *f) {
    const size_t n = f->region_id.size();
    f->t_off.resize(n);
    f->q_off.resize(n);
    f->t_cnt.resize(n);
    f->q_cnt.resize(n);
    for (size_t r = 0; r < n; ++r) {
        f->t_off[r] = f->in_off[2 * r];
        f->q_off[r] = f->in_off[2 * r + 1];
        f->t_cnt[r] = f->in_cnt[2 * r];
        f->q_cnt[r] = f->in_cnt[2 * r + 1];
    }
    memset(&f->multi, 0, sizeof(f->multi));
    avk_region_batch &b = f->batch;
    memset(&b, 0, sizeof(b));
    b.n_regions = f->region_id.size();
    b.region_id = f->region_id.data();
    b.contig_idx = f->contig_idx.data();
    b.start = f->start.data();
    b.end = f->end.data();
    b.t_off = f->t_off.data();
    b.t_cnt = f->t_cnt.data();
    b.q_off = f->q_off.data();
    b.q_cnt = f->q_cnt.data();
    b.n_variants = f->var_pos.size();
    b.var_pos = f->var_pos.data();
    b.var_type = f->var_type.data();
    b.var_zyg = f->var_zyg.data();
    b.var_raw_space = f->var_raw.data();
    b.a0_off = f->a0_off.data();
    b.a0_len = f->a0_len.data();
    b.a1_off = f->a1_off.data();
    b.a1_len = f->a1_len.data();
    b.allele_bytes = f->alleles.data();
    b.allele_bytes_len = f->alleles.size();
}

static void fill_multi_batch(avf_feed *f, uint32_t n_inputs) {
    f->is_merge = true;
    memset(&f->batch, 0, sizeof(f->batch));
    avk_multi_batch &b = f->multi;
    memset(&b, 0, sizeof(b));
    b.n_regions = f->region_id.size();
    b.n_inputs = n_inputs;
    b.region_id = f->region_id.data();
    b.contig_idx = f->contig_idx.data();
    b.start = f->start.data();
    b.end = f->end.data();
    b.in_off = f->in_off.data();
    b.in_cnt = f->in_cnt.data();
    b.n_variants = f->var_pos.size();
    b.var_pos = f->var_pos.data();
    b.var_type = f->var_type.data();
    b.var_zyg = f->var_zyg.data();
    b.var_raw_space = f->var_raw.data();
    b.a0_off = f->a0_off.data();
    b.a0_len = f->a0_len.data();
    b.a1_off = f->a1_off.data();
    b.a1_len = f->a1_len.data();
    b.allele_bytes = f->alleles.data();
    b.allele_bytes_len = f->alleles.size();
}

int avf_vcf_sample_name(const char *vcf, uint32_t index, char *out, uint64_t cap) {
    if (!vcf || !out || cap == 0) return fail(AVK_E_ARG, "null argument");
    out[0] = 0;
    LineReader in(vcf);
    if (!in.ok()) return fail(AVK_E_ARG, "Error while opening %s", vcf);
    std::string line;
    std::vector<std::string> f;
    while (in.next(line)) {
        if (line.empty() || line[0] != '#') break;
        if (line.compare(0, 6, "#CHROM") != 0) continue;
        split(line, '\t', f);
        if (f.size() <= 9 + (size_t)index) return fail(AVK_E_ARG, "Sample index %u does not exist.", index);
        const std::string &name = f[9 + index];
        if (name.size() + 1 > cap) return fail(AVK_E_ARG, "sample name of %s does not fit %llu bytes", vcf, (unsigned long long)cap);
        memcpy(out, name.c_str(), name.size() + 1);
        return 0;
    }
    return fail(AVK_E_ARG, "%s has no #CHROM header line", vcf);
}

int avf_calls_load(const char *vcf, const char *sample, int enable_trimming, avf_calls **out) {
    if (!vcf || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    avf_calls *c = new avf_calls();
    const int rc = load_vcf(vcf, sample, enable_trimming != 0, c->by_chrom);
    if (rc) {
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}
void avf_calls_free(avf_calls *c) { delete c; }
uint64_t avf_calls_count(const avf_calls *c) {
    uint64_t n = 0;
    if (c)
        for (const auto &kv : c->by_chrom)
            for (const auto &ch : kv.second.chunks) n += ch.size();
    return n;
}

int avf_feed_from_calls(uint32_t n_inputs, const avf_calls *const *calls, const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap, int merge,
                        avf_feed **out) {
    if (!calls || !g || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    if (n_inputs == 0) return fail(AVK_E_ARG, "Must provide at least 1 VCF to iterate on");
    if (n_inputs > 64) return fail(AVK_E_ARG, "at most 64 input VCFs are supported, got %u", n_inputs);
    if (!merge && n_inputs != 2) return fail(AVK_E_ARG, "a compare feed has exactly two inputs (truth, query), got %u", n_inputs);
    std::vector<const CallMap *> maps(n_inputs);
    for (uint32_t i = 0; i < n_inputs; ++i) {
        if (!calls[i]) return fail(AVK_E_ARG, "null argument");
        maps[i] = &calls[i]->by_chrom;
    }
    avf_feed *f = new avf_feed();
    const int rc = build_regions(n_inputs, maps.data(), regions_bed, g, min_variant_gap, f);
    if (rc) {
        delete f;
        return rc;
    }
    if (merge) fill_multi_batch(f, n_inputs);
    else fill_compare_batch(f);
    *out = f;
    return 0;
}

static int feed_from_files(uint32_t k, const char *const *vcfs, const char *const *samples, const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap,
                           int enable_trimming, int merge, avf_feed **out) {
    /* argument checks that come before any file is read */
    if (!regions_bed || !*regions_bed) return fail(AVK_E_ARG, "High confidence regions are currently required.");
    if (min_variant_gap == 0) return fail(AVK_E_ARG, "--min-variant-gap must be >0");
    std::vector<CallMap> calls;
    int rc = load_call_sets(k, vcfs, samples, enable_trimming, calls);
    if (rc) return rc;
    std::vector<const CallMap *> maps(k);
    for (uint32_t i = 0; i < k; ++i) maps[i] = &calls[i];
    avf_feed *f = new avf_feed();
    rc = build_regions(k, maps.data(), regions_bed, g, min_variant_gap, f);
    if (rc) {
        delete f;
        return rc;
    }
    if (merge) fill_multi_batch(f, k);
    else fill_compare_batch(f);
    *out = f;
    return 0;
}

int avf_feed_compare(const char *truth_vcf, const char *truth_sample, const char *query_vcf, const char *query_sample,
                     const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap, int enable_trimming, avf_feed **out) {
    if (!truth_vcf || !query_vcf || !g || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    const char *paths[2] = {truth_vcf, query_vcf}, *samples[2] = {truth_sample, query_sample};
    return feed_from_files(2, paths, samples, regions_bed, g, min_variant_gap, enable_trimming, 0, out);
}

int avf_feed_merge(uint32_t n_inputs, const char *const *vcfs, const char *const *samples, const char *regions_bed, const avf_genome *g,
                   uint64_t min_variant_gap, int enable_trimming, avf_feed **out) {
    if (!vcfs || !g || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    if (n_inputs == 0) return fail(AVK_E_ARG, "Must provide at least 1 VCF to iterate on");
    if (n_inputs > 64) return fail(AVK_E_ARG, "at most 64 input VCFs are supported, got %u", n_inputs);
    for (uint32_t i = 0; i < n_inputs; ++i)
        if (!vcfs[i]) return fail(AVK_E_ARG, "null argument");
    return feed_from_files(n_inputs, vcfs, samples, regions_bed, g, min_variant_gap, enable_trimming, 1, out);
}

const avk_region_batch *avf_feed_batch(const avf_feed *f) { return f && !f->is_merge ? &f->batch : nullptr; }

int avf_feed_pack(const avf_feed *f, void *(*alloc)(void *, size_t), void *user, avk_packed_batch *out) {
    if (!f || f->is_merge || !alloc || !out) return fail(AVK_E_ARG, "null argument, or a merge feed");
    const avk_region_batch &b = f->batch;
    const uint64_t n = b.n_regions, nv = b.n_variants, na = nv ? b.allele_bytes_len : 0;
    memset(out, 0, sizeof(*out));
    /* the form's constraints (aardvark_amd.h); the layout ones hold by construction (flush() above) and are checked all the same */
    if (nv >= (1ull << 32) || na >= (1ull << 32)) return 1;
    std::atomic<int> bad{0}, raw_differs{0};
    const unsigned hw = avk_usable_cpus();
    const unsigned n_threads = std::max(1u, std::min(hw ? hw : 1u, 16u));
    auto parallel = [&](uint64_t count, const std::function<void(uint64_t, uint64_t)> &fn) {
        std::vector<std::thread> pool;
        const uint64_t per = (count + n_threads - 1) / n_threads;
        for (unsigned t = 0; t < n_threads; ++t) {
            const uint64_t lo = std::min(count, t * per), hi = std::min(count, lo + per);
            if (lo < hi) pool.emplace_back(fn, lo, hi);
        }
        for (std::thread &t : pool) t.join();
    };
    parallel(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; ++r) {
            const uint64_t len = b.end[r] - b.start[r];
            const uint64_t next = r + 1 < n ? b.t_off[r + 1] : nv;
            if (b.end[r] < b.start[r] || len > 0xFFFF || b.start[r] > 0xFFFFFFFFull || b.contig_idx[r] > 0xFFFF || b.t_cnt[r] > 255 || b.q_cnt[r] > 255 ||
                b.q_off[r] != b.t_off[r] + b.t_cnt[r] || next != b.q_off[r] + b.q_cnt[r] || (r == 0 && b.t_off[0] != 0)) {
                bad = 1;
                return;
            }
            for (uint64_t v = b.t_off[r]; v < next; ++v)
                if (b.var_pos[v] < b.start[r] || b.var_pos[v] - b.start[r] > 0xFFFF) {
                    bad = 1;
                    return;
                }
        }
    });
    parallel(nv, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t v = lo; v < hi; ++v) {
            const uint64_t next = v + 1 < nv ? b.a0_off[v + 1] : na;
            if (b.a0_len[v] > 255 || b.a1_len[v] > 255 || b.var_type[v] > 15 || b.var_zyg[v] > 15 || b.a1_off[v] != b.a0_off[v] + b.a0_len[v] ||
                next != b.a1_off[v] + b.a1_len[v] || (v == 0 && b.a0_off[0] != 0)) {
                bad = 1;
                return;
            }
            if (b.var_raw_space[v] != std::max(b.a0_len[v], b.a1_len[v])) raw_differs = 1;
        }
    });
    if (bad) return 1;
    bool oom = false;
    auto get = [&](size_t bytes) {
        void *p = alloc(user, bytes ? bytes : 1);
        if (!p) oom = true;
        return p;
    };
    uint16_t *contig = (uint16_t *)get(n * 2), *len = (uint16_t *)get(n * 2), *rel = (uint16_t *)get(nv * 2);
    uint32_t *start = (uint32_t *)get(n * 4), *raw = raw_differs ? (uint32_t *)get(nv * 4) : nullptr;
    uint8_t *tc = (uint8_t *)get(n), *qc = (uint8_t *)get(n), *tz = (uint8_t *)get(nv), *l0 = (uint8_t *)get(nv), *l1 = (uint8_t *)get(nv), *bytes = (uint8_t *)get(na);
    if (oom) return fail(AVK_E_OOM, "the allocator returned NULL for an array of the packed form");
    parallel(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; ++r) {
            contig[r] = (uint16_t)b.contig_idx[r], start[r] = (uint32_t)b.start[r], len[r] = (uint16_t)(b.end[r] - b.start[r]);
            tc[r] = (uint8_t)b.t_cnt[r], qc[r] = (uint8_t)b.q_cnt[r];
            const uint64_t next = b.q_off[r] + b.q_cnt[r];
            for (uint64_t v = b.t_off[r]; v < next; ++v) rel[v] = (uint16_t)(b.var_pos[v] - b.start[r]);
        }
    });
    parallel(nv, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t v = lo; v < hi; ++v) {
            tz[v] = (uint8_t)(b.var_type[v] | b.var_zyg[v] << 4), l0[v] = (uint8_t)b.a0_len[v], l1[v] = (uint8_t)b.a1_len[v];
            if (raw) raw[v] = b.var_raw_space[v];
        }
    });
    if (na) memcpy(bytes, b.allele_bytes, na);
    out->n_regions = n, out->contig_idx = contig, out->start = start, out->len = len, out->t_cnt = tc, out->q_cnt = qc;
    out->n_variants = nv, out->var_rel_pos = rel, out->var_type_zyg = tz, out->a0_len = l0, out->a1_len = l1, out->var_raw_space = raw;
    out->allele_bytes = bytes, out->allele_bytes_len = na;
    return 0;
}

int avf_packed_slice(const avf_feed *f, const avk_packed_batch *all, uint64_t first, uint64_t n, avk_packed_batch *part, uint64_t *v_first) {
    if (!f || f->is_merge || !all || !part || !v_first) return fail(AVK_E_ARG, "null argument, or a merge feed");
    const avk_region_batch &b = f->batch;
    if (all->n_regions != b.n_regions || all->n_variants != b.n_variants) return fail(AVK_E_ARG, "the packed batch is not this feed's");
    if (first > b.n_regions || n > b.n_regions - first) return fail(AVK_E_ARG, "regions [%llu, +%llu) of %llu", (unsigned long long)first, (unsigned long long)n, (unsigned long long)b.n_regions);
    const uint64_t v0 = first < b.n_regions ? b.t_off[first] : b.n_variants, v1 = first + n < b.n_regions ? b.t_off[first + n] : b.n_variants;
    const uint64_t a0 = v0 < b.n_variants ? b.a0_off[v0] : all->allele_bytes_len, a1 = v1 < b.n_variants ? b.a0_off[v1] : all->allele_bytes_len;
    *part = *all;
    part->n_regions = n, part->contig_idx = all->contig_idx + first, part->start = all->start + first, part->len = all->len + first;
    part->t_cnt = all->t_cnt + first, part->q_cnt = all->q_cnt + first;
    part->n_variants = v1 - v0, part->var_rel_pos = all->var_rel_pos + v0, part->var_type_zyg = all->var_type_zyg + v0;
    part->a0_len = all->a0_len + v0, part->a1_len = all->a1_len + v0, part->var_raw_space = all->var_raw_space ? all->var_raw_space + v0 : nullptr;
    part->allele_bytes = all->allele_bytes + a0, part->allele_bytes_len = a1 - a0;
    *v_first = v0;
    return 0;
}
const avk_multi_batch *avf_feed_multi_batch(const avf_feed *f) { return f && f->is_merge ? &f->multi : nullptr; }

int avf_feed_pack_multi(const avf_feed *f, void *(*alloc)(void *, size_t), void *user, avk_packed_multi_batch *out) {
    if (!f || !f->is_merge || !alloc || !out) return fail(AVK_E_ARG, "null argument, or a compare feed");
    const avk_multi_batch &b = f->multi;
    const uint64_t n = b.n_regions, k = b.n_inputs, nv = b.n_variants, na = nv ? b.allele_bytes_len : 0;
    memset(out, 0, sizeof(*out));
    if (nv >= (1ull << 32) || na >= (1ull << 32)) return 1;
    bool raw_differs = false;
    for (uint64_t r = 0; r < n; ++r) { /* the form's constraints; the layout ones hold by construction (flush() above) and are checked all the same */
        const uint64_t next = r + 1 < n ? b.in_off[(r + 1) * k] : nv;
        if (b.end[r] < b.start[r] || b.end[r] - b.start[r] > 0xFFFF || b.start[r] > 0xFFFFFFFFull || b.contig_idx[r] > 0xFFFF || (r == 0 && n && b.in_off[0] != 0)) return 1;
        uint64_t at = b.in_off[r * k];
        for (uint64_t i = 0; i < k; ++i) {
            if (b.in_off[r * k + i] != at || b.in_cnt[r * k + i] > 255) return 1;
            at += b.in_cnt[r * k + i];
        }
        if (at != next) return 1;
        for (uint64_t v = b.in_off[r * k]; v < next; ++v)
            if (b.var_pos[v] < b.start[r] || b.var_pos[v] - b.start[r] > 0xFFFF) return 1;
    }
    for (uint64_t v = 0; v < nv; ++v) {
        const uint64_t next = v + 1 < nv ? b.a0_off[v + 1] : na;
        if (b.a0_len[v] > 255 || b.a1_len[v] > 255 || b.var_type[v] > 15 || b.var_zyg[v] > 15 || b.a1_off[v] != b.a0_off[v] + b.a0_len[v] || next != b.a1_off[v] + b.a1_len[v] ||
            (v == 0 && b.a0_off[0] != 0))
            return 1;
        if (b.var_raw_space[v] != std::max(b.a0_len[v], b.a1_len[v])) raw_differs = true;
    }
    bool oom = false;
    auto get = [&](size_t bytes) {
        void *p = alloc(user, bytes ? bytes : 1);
        if (!p) oom = true;
        return p;
    };
    uint16_t *contig = (uint16_t *)get(n * 2), *len = (uint16_t *)get(n * 2), *rel = (uint16_t *)get(nv * 2);
    uint32_t *start = (uint32_t *)get(n * 4), *raw = raw_differs ? (uint32_t *)get(nv * 4) : nullptr;
    uint8_t *ic = (uint8_t *)get(n * k), *tz = (uint8_t *)get(nv), *l0 = (uint8_t *)get(nv), *l1 = (uint8_t *)get(nv), *bytes = (uint8_t *)get(na);
    if (oom) return fail(AVK_E_OOM, "the allocator returned NULL for an array of the packed form");
    for (uint64_t r = 0; r < n; ++r) {
        contig[r] = (uint16_t)b.contig_idx[r], start[r] = (uint32_t)b.start[r], len[r] = (uint16_t)(b.end[r] - b.start[r]);
        for (uint64_t i = 0; i < k; ++i) ic[r * k + i] = (uint8_t)b.in_cnt[r * k + i];
        const uint64_t next = r + 1 < n ? b.in_off[(r + 1) * k] : nv;
        for (uint64_t v = b.in_off[r * k]; v < next; ++v) rel[v] = (uint16_t)(b.var_pos[v] - b.start[r]);
    }
    for (uint64_t v = 0; v < nv; ++v) {
        tz[v] = (uint8_t)(b.var_type[v] | b.var_zyg[v] << 4), l0[v] = (uint8_t)b.a0_len[v], l1[v] = (uint8_t)b.a1_len[v];
        if (raw) raw[v] = b.var_raw_space[v];
    }
    if (na) memcpy(bytes, b.allele_bytes, na);
    out->n_regions = n, out->n_inputs = (uint32_t)k, out->contig_idx = contig, out->start = start, out->len = len, out->in_cnt = ic;
    out->n_variants = nv, out->var_rel_pos = rel, out->var_type_zyg = tz, out->a0_len = l0, out->a1_len = l1, out->var_raw_space = raw;
    out->allele_bytes = bytes, out->allele_bytes_len = na;
    return 0;
}

int avf_packed_multi_slice(const avf_feed *f, const avk_packed_multi_batch *all, uint64_t first, uint64_t n, avk_packed_multi_batch *part) {
    if (!f || !f->is_merge || !all || !part) return fail(AVK_E_ARG, "null argument, or a compare feed");
    const avk_multi_batch &b = f->multi;
    const uint64_t k = b.n_inputs;
    if (all->n_regions != b.n_regions || all->n_variants != b.n_variants || all->n_inputs != b.n_inputs) return fail(AVK_E_ARG, "the packed batch is not this feed's");
    if (first > b.n_regions || n > b.n_regions - first) return fail(AVK_E_ARG, "regions [%llu, +%llu) of %llu", (unsigned long long)first, (unsigned long long)n, (unsigned long long)b.n_regions);
    const uint64_t v0 = first < b.n_regions ? b.in_off[first * k] : b.n_variants, v1 = first + n < b.n_regions ? b.in_off[(first + n) * k] : b.n_variants;
    const uint64_t a0 = v0 < b.n_variants ? b.a0_off[v0] : all->allele_bytes_len, a1 = v1 < b.n_variants ? b.a0_off[v1] : all->allele_bytes_len;
    *part = *all;
    part->n_regions = n, part->contig_idx = all->contig_idx + first, part->start = all->start + first, part->len = all->len + first, part->in_cnt = all->in_cnt + first * k;
    part->n_variants = v1 - v0, part->var_rel_pos = all->var_rel_pos + v0, part->var_type_zyg = all->var_type_zyg + v0;
    part->a0_len = all->a0_len + v0, part->a1_len = all->a1_len + v0, part->var_raw_space = all->var_raw_space ? all->var_raw_space + v0 : nullptr;
    part->allele_bytes = all->allele_bytes + a0, part->allele_bytes_len = a1 - a0;
    return 0;
}
const uint64_t *avf_feed_var_record(const avf_feed *f) { return f ? f->var_record.data() : nullptr; }
const uint32_t *avf_feed_var_alt_index(const avf_feed *f) { return f ? f->var_alt.data() : nullptr; }
uint64_t avf_feed_loaded_variants(const avf_feed *f, int input) { return f && input >= 0 && (size_t)input < f->loaded.size() ? f->loaded[input] : 0; }
void avf_feed_free(avf_feed *f) { delete f; }

int avf_write_summary_stratified(const char *path, const char *compare_label, const uint64_t *tally, const avf_strat *strat, const uint64_t *strat_tallies,
                                 uint32_t metrics_mask) {
    if (!path || !tally || (strat && !strat_tallies)) return fail(AVK_E_ARG, "null argument");
    const std::string p(path);
    const char delim = p.size() >= 4 && p.compare(p.size() - 4, 4, ".csv") == 0 ? ',' : '\t';
    FILE *fp = fopen(path, "w");
    if (!fp) return fail(AVK_E_ARG, "cannot create %s", path);
    const std::string label = csv_field(compare_label ? compare_label : "", delim);
    static const char *const header[16] = {"compare_label", "comparison", "region_label", "filter", "variant_type", "truth_total", "truth_tp", "truth_fn",
                                           "query_total", "query_tp", "query_fp", "metric_recall", "metric_precision", "metric_f1", "truth_fn_gt", "query_fp_gt"};
    for (int k = 0; k < 16; ++k) fprintf(fp, "%s%c", header[k], k == 15 ? '\n' : delim);
    /* Debug names of VariantType in declaration order = BTreeMap iteration order (variants.rs:6-31) */
    static const char *const type_name[12] = {"Snv", "Insertion", "Deletion", "Indel", "SvInsertion", "SvDeletion", "SvDuplication", "SvInversion",
                                              "SvBreakend", "TrContraction", "TrExpansion", "Unknown"};
    struct Joint {
        const char *label;
        std::vector<int> types;
    };
    const Joint joints[3] = {{"JointIndel", {AVK_VT_INSERTION, AVK_VT_DELETION, AVK_VT_INDEL}},
                             {"JointStructuralVariant", {AVK_VT_SV_INSERTION, AVK_VT_SV_DELETION, AVK_VT_SV_DUPLICATION, AVK_VT_SV_INVERSION, AVK_VT_SV_BREAKEND}},
                             {"JointTandemRepeat", {AVK_VT_TR_EXPANSION, AVK_VT_TR_CONTRACTION}}};
    /* metric kinds in the order main.rs pushes them (:134-147): GT, BASEPAIR, HAP, WEIGHTED_HAP, RECORD_BP */
    struct Kind {
        uint32_t bit;
        const char *name;
        int base; /* first of the 4 fields truth_tp, truth_fn, query_tp, query_fp */
        bool gt;
    };
    const Kind kinds[5] = {{AVF_METRIC_GT, "GT", AVK_F_GT_TRUTH_TP, true},
                           {AVF_METRIC_BASEPAIR, "BASEPAIR", AVK_F_BP_TRUTH_TP, false},
                           {AVF_METRIC_HAP, "HAP", AVK_F_HAP_TRUTH_TP, false},
                           {AVF_METRIC_WEIGHTED_HAP, "WEIGHTED_HAP", AVK_F_WHAP_TRUTH_TP, false},
                           {AVF_METRIC_RECORD_BP, "RECORD_BP", AVK_F_RBP_TRUTH_TP, false}};
    std::string region_label = "ALL";
    auto row = [&](const Kind &kd, const char *vtype, const uint64_t m[4], const uint64_t gt_extra[2]) {
        const uint64_t ttot = m[0] + m[1], qtot = m[2] + m[3];
        std::string recall, precision, f1;
        if (ttot > 0) recall = fmt_f64((double)m[0] / (double)ttot);
        if (qtot > 0) precision = fmt_f64((double)m[2] / (double)qtot);
        if (ttot > 0 && qtot > 0) {
            const double r = (double)m[0] / (double)ttot, pr = (double)m[2] / (double)qtot;
            f1 = fmt_f64(2.0 * r * pr / (r + pr));
        }
        fprintf(fp, "%s%c%s%c%s%cALL%c%s%c%llu%c%llu%c%llu%c%llu%c%llu%c%llu%c%s%c%s%c%s%c", label.c_str(), delim, kd.name, delim, region_label.c_str(), delim, delim, vtype, delim,
                (unsigned long long)ttot, delim, (unsigned long long)m[0], delim, (unsigned long long)m[1], delim, (unsigned long long)qtot, delim,
                (unsigned long long)m[2], delim, (unsigned long long)m[3], delim, recall.c_str(), delim, precision.c_str(), delim, f1.c_str(), delim);
        if (kd.gt) fprintf(fp, "%llu%c%llu\n", (unsigned long long)gt_extra[0], delim, (unsigned long long)gt_extra[1]);
        else fprintf(fp, "%c\n", delim);
    };
    auto write_group = [&](const uint64_t *tally) {
        for (const Kind &kd : kinds) {
            if (!(metrics_mask & kd.bit)) continue;
            auto fields = [&](int group, uint64_t m[4], uint64_t ex[2]) {
                const uint64_t *gp = tally + (size_t)group * AVK_N_FIELDS;
                for (int k = 0; k < 4; ++k) m[k] = gp[kd.base + k];
                ex[0] = gp[AVK_F_GT_TRUTH_FN_GT];
                ex[1] = gp[AVK_F_GT_QUERY_FP_GT];
            };
            uint64_t m[4], ex[2];
            fields(0, m, ex);
            row(kd, "ALL", m, ex);
            for (int t = 0; t < AVK_N_VARIANT_TYPES; ++t) {
                fields(1 + t, m, ex);
                if (m[0] + m[1] + m[2] + m[3] == 0) continue; /* is_empty: no row */
                row(kd, type_name[t], m, ex);
            }
            for (const Joint &jt : joints) {
                uint64_t s[4] = {0, 0, 0, 0}, sx[2] = {0, 0};
                for (int t : jt.types) {
                    fields(1 + t, m, ex);
                    for (int k = 0; k < 4; ++k) s[k] += m[k];
                    sx[0] += ex[0];
                    sx[1] += ex[1];
                }
                if (s[0] + s[1] + s[2] + s[3] == 0) continue;
                row(kd, jt.label, s, sx);
            }
        }
    };
    write_group(tally); /* the ALL group first (summary.rs:196-201) */
    if (strat)
        for (uint32_t l = 0; l < avf_strat_n_labels(strat); ++l) {
            region_label = csv_field(avf_strat_label(strat, l), delim);
            write_group(strat_tallies + (size_t)l * AVK_TALLY_LEN);
        }
    const bool bad = ferror(fp) != 0;
    if (fclose(fp) != 0 || bad) return fail(AVK_E_ARG, "write error on %s", path);
    return 0;
}


int avf_write_summary(const char *path, const char *compare_label, const uint64_t *tally, uint32_t metrics_mask) {
    return avf_write_summary_stratified(path, compare_label, tally, nullptr, nullptr, metrics_mask);
}

} /* extern "C" */
