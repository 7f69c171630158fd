/*
 * avf_feeder.cpp — libaardvark_feeder.so: FASTA / BED / VCF readers, the reference's region generation and its
 * summary writer, restated in C++ behind the C-ABI of include/aardvark_feeder.h.  Host code only.
 * file:line citations are into PacificBiosciences/aardvark v0.10.5.
 */
#include "../../../include/aardvark_feeder.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
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

bool parse_u64(const std::string &s, uint64_t &v) {
    if (s.empty()) return false;
    const auto r = std::from_chars(s.data(), s.data() + s.size(), v);
    return r.ec == std::errc() && r.ptr == s.data() + s.size();
}

} // namespace

/* ------------------------------------------------------------------------------------------ genome */
struct avf_genome {
    std::vector<std::string> names;
    std::vector<std::vector<uint8_t>> seqs;
    std::unordered_map<std::string, uint32_t> index;
};

/* ------------------------------------------------------------------------------------------ feed */
struct avf_feed {
    uint32_t k = 2;              /* inputs: 2 for compare (truth, query), the number of VCFs for merge */
    bool is_merge = false;
    avk_region_batch batch;      /* compare feeds */
    avk_multi_batch multi;       /* merge feeds */
    std::vector<uint64_t> region_id, start, end, in_off, t_off, q_off, var_pos, a0_off, a1_off, var_record;
    std::vector<uint32_t> contig_idx, in_cnt, t_cnt, q_cnt, var_raw, a0_len, a1_len, var_alt;
    std::vector<uint8_t> var_type, var_zyg, alleles;
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
int load_vcf(const char *path, const char *sample, bool enable_trimming, std::unordered_map<std::string, std::vector<Call>> &calls) {
    LineReader in(path);
    if (!in.ok()) return fail(AVK_E_ARG, "Error while opening %s", path);
    std::string line;
    std::vector<std::string> f, alts, fmt, sv, info;
    long sample_col = -1;
    uint64_t record = 0;
    bool have_header = false;
    while (in.next(line)) {
        if (line.empty()) continue;
        if (line[0] == '#') {
            if (line.compare(0, 6, "#CHROM") == 0) {
                split(line, '\t', f);
                have_header = true;
                if (f.size() < 10) return fail(AVK_E_ARG, "%s has no sample columns", path);
                if (!sample || !*sample) sample_col = 9; /* get_vcf_sample_name(.., 0) */
                else {
                    for (size_t k = 9; k < f.size(); ++k)
                        if (f[k] == sample) {
                            sample_col = (long)k;
                            break;
                        }
                    if (sample_col < 0) return fail(AVK_E_ARG, "Sample name \"%s\" was not found in %s", sample, path);
                }
            }
            continue;
        }
        if (!have_header) return fail(AVK_E_ARG, "%s: data line before the #CHROM header", path);
        const uint64_t rec = record++;
        split(line, '\t', f);
        if ((long)f.size() <= sample_col) return fail(AVK_E_ARG, "%s: record %llu has too few columns", path, (unsigned long long)rec);
        uint64_t pos1 = 0;
        if (!parse_u64(f[1], pos1) || pos1 == 0) return fail(AVK_E_ARG, "%s: record %llu: Missing POS", path, (unsigned long long)rec);
        const std::string &ref_seq = f[3];
        /* sample GT */
        split(f[8], ':', fmt);
        long gt_at = -1;
        for (size_t k = 0; k < fmt.size(); ++k)
            if (fmt[k] == "GT") {
                gt_at = (long)k;
                break;
            }
        if (gt_at < 0) return fail(AVK_E_ARG, "%s: record %llu (%s:%llu): Missing GT", path, (unsigned long long)rec, f[0].c_str(), (unsigned long long)pos1);
        split(f[(size_t)sample_col], ':', sv);
        if ((long)sv.size() <= gt_at || sv[(size_t)gt_at] == "." || sv[(size_t)gt_at].empty()) continue; /* GT = '.': a no-op (:583-586) */
        const std::string &gt = sv[(size_t)gt_at];
        /* parse_genotype (:660-712) */
        uint64_t idx[2] = {0, 0};
        bool phased = false;
        {
            size_t n_alleles = 0, b = 0;
            for (size_t k = 0; k <= gt.size(); ++k) {
                if (k == gt.size() || gt[k] == '/' || gt[k] == '|') {
                    if (k < gt.size() && gt[k] == '|') phased = true;
                    if (n_alleles >= 2) return fail(AVK_E_ARG, "%s: record %llu: allele.len() != [1, 2]: %s", path, (unsigned long long)rec, gt.c_str());
                    const std::string a(gt, b, k - b);
                    uint64_t v = 0;
                    if (a != "." && !parse_u64(a, v)) return fail(AVK_E_ARG, "%s: record %llu: malformed GT %s", path, (unsigned long long)rec, gt.c_str());
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
        if (n_picks == 0) continue;
        if (f[4] == "." || f[4].empty()) alts.clear();
        else split(f[4], ',', alts);
        /* INFO: SVTYPE and TRID */
        bool has_svtype = false, has_trid = false;
        std::string svtype;
        if (f[7] != "." && !f[7].empty()) {
            split(f[7], ';', info);
            for (const std::string &kv : info) {
                if (kv.compare(0, 7, "SVTYPE=") == 0) {
                    has_svtype = true;
                    svtype.assign(kv, 7);
                } else if (kv.compare(0, 5, "TRID=") == 0 && kv.size() > 5) has_trid = true;
            }
        }
        for (int p = 0; p < n_picks; ++p) {
            const uint64_t alt_index = picks[p].first;
            if (alt_index > alts.size()) return fail(AVK_E_ARG, "%s: record %llu: GT refers to ALT %llu of %zu", path, (unsigned long long)rec, (unsigned long long)alt_index, alts.size());
            const std::string &alt = alts[alt_index - 1];
            if (alt == "*") continue;            /* effectively a reference allele (:597-600) */
            if (!alt.empty() && alt[0] == '<') continue; /* symbolic: needs sequence-resolved (:604-607) */
            std::string r = ref_seq, a = alt;
            const size_t raw_space = std::max(r.size(), a.size()); /* before trimming (:612) */
            while (enable_trimming && r.size() > 1 && a.size() > 1 && r.back() == a.back()) {
                r.pop_back();
                a.pop_back();
            }
            if (r.size() > 10000 || a.size() > 10000) continue; /* allele_size_limit (:621-626) */
            uint8_t type = 0;
            const int rc = variant_type_of(svtype, has_svtype, has_trid, r.size(), a.size(), type);
            if (rc == 1) continue;
            if (rc < 0) return fail(AVK_E_ARG, "%s: record %llu (%s:%llu): %s", path, (unsigned long long)rec, f[0].c_str(), (unsigned long long)pos1, std::string(t_error).c_str());
            Call c;
            c.pos = pos1 - 1;
            c.a0.swap(r);
            c.a1.swap(a);
            c.raw_space = (uint32_t)raw_space;
            c.type = type;
            c.zyg = picks[p].second;
            c.record = rec;
            c.alt_index = (uint32_t)alt_index;
            calls[f[0]].push_back(std::move(c));
        }
    }
    if (in.failed()) return fail(AVK_E_ARG, "read error in %s", path);
    if (!have_header) return fail(AVK_E_ARG, "%s has no #CHROM header line", path);
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

int avf_genome_load(const char *fasta_path, avf_genome **out) {
    if (!fasta_path || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    LineReader in(fasta_path);
    if (!in.ok()) return fail(AVK_E_ARG, "cannot open FASTA file %s", fasta_path);
    avf_genome *g = new avf_genome();
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
static int build_regions(uint32_t k, const char *const *vcfs, const char *const *samples, const char *regions_bed, const avf_genome *g,
                         uint64_t min_variant_gap, int enable_trimming, avf_feed *f) {
    if (!regions_bed || !*regions_bed) return fail(AVK_E_ARG, "High confidence regions are currently required.");
    if (min_variant_gap == 0) return fail(AVK_E_ARG, "--min-variant-gap must be >0");
    LoadedBed bed;
    int rc = load_bed(regions_bed, bed);
    if (rc) return rc;
    /* the files are read side by side, like the reference's per-file parallel load (:324-347) */
    std::vector<std::unordered_map<std::string, std::vector<Call>>> calls(k);
    std::vector<int> rcs(k, 0);
    std::vector<std::string> errs(k);
    {
        const uint32_t n_threads = std::min<uint32_t>(k, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
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

    f->k = k;
    f->loaded.assign(k, 0);
    struct Joint {
        uint32_t input;
        const Call *c;
    };
    std::vector<Joint> joint;
    uint64_t next_region_id = 0;
    std::vector<std::vector<const Call *>> vars(k);
    for (size_t ci = 0; ci < bed.chroms.size(); ++ci) {
        const std::string &chrom = bed.chroms[ci];
        const std::vector<Interval1> &intervals = bed.intervals[ci];
        const auto git = g->index.find(chrom);
        if (git == g->index.end()) return fail(AVK_E_ARG, "Chromosome %s was not found in reference genome", chrom.c_str());
        const uint32_t contig = git->second;
        const uint64_t chrom_length = g->seqs[contig].size();
        /* the span the reference queries through tabix: first interval's start to the LAST interval's end (:289-296) */
        const uint64_t zb_start = intervals.front().start - 1, zb_end = intervals.back().end;
        joint.clear();
        for (uint32_t input = 0; input < k; ++input) {
            const auto it = calls[input].find(chrom);
            if (it == calls[input].end()) continue;
            for (const Call &c : it->second) {
                /* is_variant_contained (:764-778): first and last reference base inside the span */
                const uint64_t last = c.pos + c.a0.size() - 1;
                if (c.pos >= zb_start && c.pos < zb_end && last >= zb_start && last < zb_end) {
                    joint.push_back(Joint{input, &c});
                    f->loaded[input] += 1;
                }
            }
        }
        std::stable_sort(joint.begin(), joint.end(), [](const Joint &a, const Joint &b) { return a.c->pos < b.c->pos; }); /* sort_by_key(position) */
        size_t head = 0; /* the deque's front */
        auto flush = [&](uint64_t ws, uint64_t we) {
            f->region_id.push_back(next_region_id++);
            f->contig_idx.push_back(contig);
            f->start.push_back(ws);
            f->end.push_back(we);
            for (uint32_t input = 0; input < k; ++input) {
                f->in_off.push_back(f->var_pos.size());
                f->in_cnt.push_back((uint32_t)vars[input].size());
                for (const Call *c : vars[input]) {
                    f->var_pos.push_back(c->pos);
                    f->var_type.push_back(c->type);
                    f->var_zyg.push_back(c->zyg);
                    f->var_raw.push_back(c->raw_space);
                    f->a0_off.push_back(f->alleles.size());
                    f->a0_len.push_back((uint32_t)c->a0.size());
                    f->alleles.insert(f->alleles.end(), c->a0.begin(), c->a0.end());
                    f->a1_off.push_back(f->alleles.size());
                    f->a1_len.push_back((uint32_t)c->a1.size());
                    f->alleles.insert(f->alleles.end(), c->a1.begin(), c->a1.end());
                    f->var_record.push_back(c->record);
                    f->var_alt.push_back(c->alt_index);
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
                vars[j.input].push_back(j.c);
            }
            if (have_window && have_end) flush(window_start, window_end);
        }
    }
    if (f->alleles.empty()) f->alleles.push_back(0);
    return 0;
}

int avf_feed_compare(const char *truth_vcf, const char *truth_sample, const char *query_vcf, const char *query_sample,
                     const char *regions_bed, const avf_genome *g, uint64_t min_variant_gap, int enable_trimming, avf_feed **out) {
    if (!truth_vcf || !query_vcf || !g || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    avf_feed *f = new avf_feed();
    const char *paths[2] = {truth_vcf, query_vcf}, *samples[2] = {truth_sample, query_sample};
    const int rc = build_regions(2, paths, samples, regions_bed, g, min_variant_gap, enable_trimming, f);
    if (rc) {
        delete f;
        return rc;
    }
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
    *out = f;
    return 0;
}

int avf_feed_merge(uint32_t n_inputs, const char *const *vcfs, const char *const *samples, const char *regions_bed, const avf_genome *g,
                   uint64_t min_variant_gap, int enable_trimming, avf_feed **out) {
    if (!vcfs || !g || !out) return fail(AVK_E_ARG, "null argument");
    *out = nullptr;
    if (n_inputs == 0) return fail(AVK_E_ARG, "Must provide at least 1 VCF to iterate on");
    if (n_inputs > 64) return fail(AVK_E_ARG, "at most 64 input VCFs are supported, got %u", n_inputs);
    for (uint32_t i = 0; i < n_inputs; ++i)
        if (!vcfs[i]) return fail(AVK_E_ARG, "null argument");
    avf_feed *f = new avf_feed();
    f->is_merge = true;
    const int rc = build_regions(n_inputs, vcfs, samples, regions_bed, g, min_variant_gap, enable_trimming, f);
    if (rc) {
        delete f;
        return rc;
    }
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
    *out = f;
    return 0;
}

const avk_region_batch *avf_feed_batch(const avf_feed *f) { return f && !f->is_merge ? &f->batch : nullptr; }
const avk_multi_batch *avf_feed_multi_batch(const avf_feed *f) { return f && f->is_merge ? &f->multi : nullptr; }
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
