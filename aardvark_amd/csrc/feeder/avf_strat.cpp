/*
 * avf_strat.cpp — Stratifications (src/parsing/stratifications.rs): labelled BED sets and the containment / overlap
 * queries solve_compare_region makes against them.  Part of libaardvark_feeder.so; host code only.
 * The reference keeps one interval tree per label and chromosome; the two queries it makes only need
 * "is there an interval with start <= x whose end reaches y", which sorted starts + a running maximum of the ends answer
 * with one binary search.
 */
#include "../../../include/aardvark_feeder.h"

#include "../avk_cpus.h"

#include <zlib.h>

#include <algorithm>
#include <map>
#include <functional>
#include <thread>
#include <string>
#include <vector>

int avf_fail_(int code, const char *fmt, ...);

struct avf_strat {
    struct Tree {
        std::vector<int64_t> start;   /* sorted */
        std::vector<int64_t> max_end; /* max_end[k] = max end over intervals 0..k */
    };
    std::vector<std::string> labels;                  /* sorted (BTreeMap order, stratifications.rs:37-66) */
    std::vector<std::map<std::string, Tree>> trees;   /* per label: chromosome -> intervals, 0-based inclusive */
};

namespace {

bool read_lines(const char *path, std::vector<std::string> &lines) {
    gzFile in = gzopen(path, "rb");
    if (!in) return false;
    std::string cur;
    char buf[1 << 16];
    int n;
    while ((n = gzread(in, buf, sizeof(buf))) > 0) {
        for (int k = 0; k < n; ++k) {
            if (buf[k] == '\n') {
                if (!cur.empty() && cur.back() == '\r') cur.pop_back();
                lines.push_back(cur);
                cur.clear();
            } else cur.push_back(buf[k]);
        }
    }
    if (!cur.empty()) lines.push_back(cur);
    gzclose(in);
    return n == 0;
}

/* true when some interval has start <= a and end >= b */
bool reaches(const avf_strat::Tree &t, int64_t a, int64_t b) {
    const size_t k = (size_t)(std::upper_bound(t.start.begin(), t.start.end(), a) - t.start.begin());
    return k > 0 && t.max_end[k - 1] >= b;
}

uint32_t query(const avf_strat *s, const char *chrom, int64_t first, int64_t last, bool contain, uint32_t *out, uint32_t cap) {
    if (!s || !chrom) return 0;
    uint32_t n = 0;
    for (uint32_t l = 0; l < s->labels.size(); ++l) {
        const auto it = s->trees[l].find(chrom);
        if (it == s->trees[l].end()) continue;
        /* contained: i.first <= first && i.last >= last (:189-199); overlapping: i.first <= last && i.last >= first (:179-186) */
        const bool hit = contain ? reaches(it->second, first, last) : reaches(it->second, last, first);
        if (hit) {
            if (out && n < cap) out[n] = l;
            n += 1;
        }
    }
    return n;
}

} // namespace

extern "C" {

int avf_strat_load(const char *tsv_path, avf_strat **out) {
    if (!tsv_path || !out) return avf_fail_(AVK_E_ARG, "null argument");
    *out = nullptr;
    std::vector<std::string> rows;
    if (!read_lines(tsv_path, rows)) return avf_fail_(AVK_E_ARG, "Error while opening %s", tsv_path);
    std::string folder(tsv_path);
    const size_t slash = folder.find_last_of('/');
    folder = slash == std::string::npos ? std::string() : folder.substr(0, slash + 1);
    std::map<std::string, std::string> files; /* label -> path, sorted like the reference's BTreeMap */
    for (const std::string &row : rows) {
        if (row.empty()) continue;
        const size_t tab = row.find('\t');
        if (tab == std::string::npos) return avf_fail_(AVK_E_ARG, "Missing filename on row: %s", row.c_str());
        const std::string label = row.substr(0, tab);
        std::string file = row.substr(tab + 1);
        const size_t tab2 = file.find('\t');
        if (tab2 != std::string::npos) file.resize(tab2);
        if (files.count(label)) return avf_fail_(AVK_E_ARG, "Duplicate label found: %s", label.c_str());
        files[label] = (!file.empty() && file[0] == '/') ? file : folder + file;
    }
    avf_strat *s = new avf_strat();
    for (const auto &kv : files) {
        std::vector<std::string> bed;
        if (!read_lines(kv.second.c_str(), bed)) {
            delete s;
            return avf_fail_(AVK_E_ARG, "Error while loading %s", kv.second.c_str());
        }
        std::map<std::string, std::vector<std::pair<int64_t, int64_t>>> by_chrom;
        for (const std::string &line : bed) {
            if (line.empty() || line[0] == '#' || line.compare(0, 5, "track") == 0 || line.compare(0, 7, "browser") == 0) continue;
            const size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1);
            if (t2 == std::string::npos) {
                delete s;
                return avf_fail_(AVK_E_ARG, "%s: malformed BED record", kv.second.c_str());
            }
            const size_t t3 = line.find('\t', t2 + 1);
            const long long b = atoll(line.substr(t1 + 1, t2 - t1 - 1).c_str());
            const long long e = atoll(line.substr(t2 + 1, t3 == std::string::npos ? std::string::npos : t3 - t2 - 1).c_str());
            by_chrom[line.substr(0, t1)].emplace_back((int64_t)b, (int64_t)e - 1); /* BED [b, e) -> 0-based inclusive [b, e-1] (:155-160) */
        }
        std::map<std::string, avf_strat::Tree> trees;
        for (auto &c : by_chrom) {
            std::sort(c.second.begin(), c.second.end());
            avf_strat::Tree t;
            int64_t m = INT64_MIN;
            for (const auto &iv : c.second) {
                t.start.push_back(iv.first);
                m = std::max(m, iv.second);
                t.max_end.push_back(m);
            }
            trees.emplace(c.first, std::move(t));
        }
        s->labels.push_back(kv.first);
        s->trees.push_back(std::move(trees));
    }
    *out = s;
    return 0;
}

uint32_t avf_strat_n_labels(const avf_strat *s) { return s ? (uint32_t)s->labels.size() : 0; }
const char *avf_strat_label(const avf_strat *s, uint32_t label) { return s && label < s->labels.size() ? s->labels[label].c_str() : nullptr; }
uint64_t avf_strat_n_intervals(const avf_strat *s, uint32_t label, const char *chrom) {
    if (!s || !chrom || label >= s->labels.size()) return 0;
    const auto it = s->trees[label].find(chrom);
    return it == s->trees[label].end() ? 0 : it->second.start.size();
}
uint32_t avf_strat_containments(const avf_strat *s, const char *chrom, int64_t first, int64_t last, uint32_t *out, uint32_t cap) {
    return query(s, chrom, first, last, true, out, cap);
}
uint32_t avf_strat_overlaps(const avf_strat *s, const char *chrom, int64_t first, int64_t last, uint32_t *out, uint32_t cap) {
    return query(s, chrom, first, last, false, out, cap);
}
uint32_t avf_strat_region_labels(const avf_strat *s, const avf_genome *g, const avk_region_batch *b, uint64_t r, uint32_t *out, uint32_t cap) {
    if (!s || !g || !b || r >= b->n_regions) return 0;
    /* CompareRegion::var_coordinates (compare_region.rs:54-66): first variants' positions, LAST variants' ends */
    uint64_t start = UINT64_MAX, end = 0;
    if (b->t_cnt[r]) {
        const uint64_t f = b->t_off[r], l = f + b->t_cnt[r] - 1;
        start = std::min(start, b->var_pos[f]);
        end = std::max(end, b->var_pos[l] + b->a0_len[l]);
    }
    if (b->q_cnt[r]) {
        const uint64_t f = b->q_off[r], l = f + b->q_cnt[r] - 1;
        start = std::min(start, b->var_pos[f]);
        end = std::max(end, b->var_pos[l] + b->a0_len[l]);
    }
    if (start >= end) return 0;
    const char *chrom = avf_genome_name(g, b->contig_idx ? b->contig_idx[r] : 0);
    return query(s, chrom, (int64_t)start, (int64_t)end - 1, true, out, cap);
}
/* the same for regions first .. first + n - 1 at once, by several threads: label_off[k] .. label_off[k + 1] delimits the labels of
 * region first + k in label_idx.  Call with label_idx = NULL to get the offsets (label_off[n] = entries needed), then again with room. */
int avf_strat_batch_labels(const avf_strat *s, const avf_genome *g, const avk_region_batch *b, uint64_t first, uint64_t n, uint64_t *label_off,
                           uint32_t *label_idx) {
    if (!s || !g || !b || !label_off || first > b->n_regions || n > b->n_regions - first) return AVK_E_ARG;
    const uint32_t n_labels = avf_strat_n_labels(s);
    unsigned hw = avk_usable_cpus();
    size_t nt = hw < 1 ? 1 : (hw > 16 ? 16 : hw);
    if (nt > n / 4096 + 1) nt = (size_t)(n / 4096 + 1);
    auto run = [&](const std::function<void(uint64_t, uint64_t)> &fn) {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(fn, n * t / nt, n * (t + 1) / nt);
        fn(0, n / nt);
        for (std::thread &t : pool) t.join();
    };
    if (!label_idx) {
        run([&](uint64_t lo, uint64_t hi) {
            std::vector<uint32_t> tmp(n_labels ? n_labels : 1);
            for (uint64_t k = lo; k < hi; ++k) label_off[k + 1] = avf_strat_region_labels(s, g, b, first + k, tmp.data(), n_labels);
        });
        label_off[0] = 0;
        for (uint64_t k = 0; k < n; ++k) label_off[k + 1] += label_off[k];
        return 0;
    }
    run([&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; ++k)
            (void)avf_strat_region_labels(s, g, b, first + k, label_idx + label_off[k], (uint32_t)(label_off[k + 1] - label_off[k]));
    });
    return 0;
}
void avf_strat_free(avf_strat *s) { delete s; }

} /* extern "C" */
