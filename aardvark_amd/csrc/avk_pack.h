/*
 * avk_pack.h — host side of the boundary: validates a caller batch (avk_region_batch) and lays
 * it out the way the kernels read it (avk_dev_types.h).  Plain C++, no HIP.
 *
 * Validation mirrors the conditions under which the reference panics or is undefined before it
 * reaches the solver (slice out of bounds in get_full_chromosome/get_slice, waffle_solver.rs:131-140,
 * :757,:773; unsorted variants, query_optimizer.rs:370-371 "pre-sorted"; empty alleles,
 * variants.rs:103-355; raw_allele_space < allele length, variants.rs:364-370; non-ALT zygosities hit
 * assert_eq! at query_optimizer.rs:315): those regions get a status instead of aborting the process.
 */
#ifndef AVK_PACK_H
#define AVK_PACK_H

#include "avk_cpus.h"

#include <cstring>
#include <string>
#include <functional>
#include <thread>
#include <vector>

#include "avk_dev_types.h"
#include "avk_pairs.inl"

namespace avk {

/* the 8 variant types add_basepair_stats filters by, in the reference's order (waffle_solver.rs:383) */
static const uint8_t AVK_SUP_TYPES[8] = {AVK_VT_SNV, AVK_VT_INSERTION, AVK_VT_DELETION, AVK_VT_INDEL,
                                         AVK_VT_TR_CONTRACTION, AVK_VT_TR_EXPANSION, AVK_VT_SV_DELETION, AVK_VT_SV_INSERTION};

/* vectors of plain records whose resize() leaves new elements uninitialised: the packer's threads write every byte that is read */
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind {
        using other = NoInitAlloc<U>;
    };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using PodVec = std::vector<T, NoInitAlloc<T>>;

/* What a lane will spend on a region, as far as the host can tell without solving it: a tile of 64 lanes takes as long as its slowest
 * region, so the records of a lane class are sorted by this key and a tile holds regions of one cost.  Almost all of the time beyond
 * the common case goes into edits (a wavefront alignment costs the square of its distance): calls without an identical call on the other
 * side (the edits of a missed or extra call, twice for a homozygous one), identical calls with different genotypes, and — the metrics
 * phase — a side whose calls have different types (its per-type alignments see the side's longest allele difference).  Measured on the
 * whole-genome workload with the emulator's work counters: sum over tiles of the slowest lane / mean lane = 2.9 (one call per side),
 * 5.4 (two), 8.2 (three) when sorted by call count only, 1.2 / 2.8 / 4.1 with this key.
 * Layout: estimated edits (capped at 15) << 4 | calls above the class's minimum (capped at 3) << 2 | calls that are not homozygous (capped at 3);
 * larger = more expensive. */
struct FastCall {
    uint32_t pos, a0, a1, alt_ed, type, zyg;
    const uint8_t *alt;
};
inline uint8_t fast_cost_key(const FastCall *t, uint32_t tc, const FastCall *q, uint32_t qc) {
    uint32_t est = 0, used = 0, nhet = 0;
    auto copies = [](uint32_t z) { return z == AVK_ZYG_HOM_ALT ? 2u : 1u; };
    for (uint32_t i = 0; i < tc; ++i) {
        int m = -1;
        for (uint32_t j = 0; j < qc && m < 0; ++j)
            if (!((used >> j) & 1u) && t[i].pos == q[j].pos && t[i].a0 == q[j].a0 && t[i].a1 == q[j].a1 && memcmp(t[i].alt, q[j].alt, t[i].a1) == 0) m = (int)j;
        if (m < 0) est += t[i].alt_ed * copies(t[i].zyg);
        else {
            used |= 1u << m;
            const uint32_t a = copies(t[i].zyg), b = copies(q[m].zyg);
            est += t[i].alt_ed * (a > b ? a - b : b - a);
        }
        nhet += t[i].zyg != AVK_ZYG_HOM_ALT;
    }
    for (uint32_t j = 0; j < qc; ++j) {
        if (!((used >> j) & 1u)) est += q[j].alt_ed * copies(q[j].zyg);
        nhet += q[j].zyg != AVK_ZYG_HOM_ALT;
    }
    for (int side = 0; side < 2; ++side) {
        const FastCall *c = side ? q : t;
        const uint32_t n = side ? qc : tc;
        bool mixed = false;
        uint32_t longest = 0;
        for (uint32_t i = 0; i < n; ++i) {
            mixed = mixed || c[i].type != c[0].type;
            longest = c[i].alt_ed > longest ? c[i].alt_ed : longest;
        }
        if (mixed && longest > 2) est += longest - 2;
    }
    const uint32_t n_all = tc + qc, lo = tc > qc ? tc : qc; /* the class's fewest calls: one side full */
    const uint32_t extra = n_all - lo;
    return (uint8_t)(((est > 15 ? 15u : est) << 4) | ((extra > 3 ? 3u : extra) << 2) | (nhet > 3 ? 3u : nhet));
}

struct PackedBatch {
    PodVec<AvkDevRegion> regions;
    PodVec<uint32_t> blob;          /* what the device reads: one blob per region (AvkBlobVar in avk_dev_types.h) */
    PodVec<AvkDevVariant> variants; /* host-side only */
    std::vector<uint8_t> alleles;        /* host-side only */
    PodVec<uint64_t> dev2host; /* device variant index -> caller variant index */
    std::vector<uint8_t> zyg_flags; /* per region: bit 0 an Unknown zygosity, bit 1 a HomozygousReference one */
    std::vector<uint8_t> nhet_u;    /* per region: unphased heterozygous calls, both sides (AVK_HET_SEARCH_MIN) */
    std::vector<int64_t> delta_t, delta_q; /* variant_delta_length per side (merge_solver.rs:211-223) */
    std::vector<uint8_t> fast_class;       /* per region: 0, or 1 + index into AVK_FAST_CLASS (eligible for the lane-per-region kernel) */
    std::vector<uint8_t> fast_key;         /* per region of a fast class: fast_cost_key (tiles hold regions of one cost) */
    uint64_t seq_total = 0;
};

inline uint32_t seq_stride_of(const avk_region_batch *b, uint64_t r) {
    uint64_t L = b->end[r] >= b->start[r] ? b->end[r] - b->start[r] : 0;
    uint64_t g[2] = {0, 0};
    for (int side = 0; side < 2; ++side) {
        uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
        uint32_t cnt = side == 0 ? b->t_cnt[r] : b->q_cnt[r];
        if (off > b->n_variants || (uint64_t)cnt > b->n_variants - off) return 1; /* the packer rejects the batch */
        for (uint32_t i = 0; i < cnt; ++i) {
            uint64_t v = off + i;
            if (b->a1_len[v] > b->a0_len[v]) g[side] += b->a1_len[v] - b->a0_len[v];
        }
    }
    uint64_t s = L + (g[0] > g[1] ? g[0] : g[1]);
    if (s < 1) s = 1;
    if (s > 0xFFFFFFFFull) s = 0xFFFFFFFFull;
    return (uint32_t)s;
}

/* Unit-cost edit distance of two byte strings = the reference's wfa_ed on complete strings
 * (src/util/sequence_alignment.rs:9-13, asserted equal to the DP edit distance at :58-116).
 * Common prefix and suffix are stripped first (every normalised SNV / insertion / deletion ends there);
 * what is left runs the furthest-reaching wavefront recurrence, O((n + m) * d). */
inline uint64_t host_edit_distance(const uint8_t *a, uint64_t n, const uint8_t *b, uint64_t m) {
    while (n && m && a[0] == b[0]) {
        ++a;
        ++b;
        --n;
        --m;
    }
    while (n && m && a[n - 1] == b[m - 1]) {
        --n;
        --m;
    }
    if (n == 0) return m;
    if (m == 0) return n;
    if (n == 1 && m == 1) return 1;
    /* fr[k + d] = furthest x (symbols of a consumed) on diagonal k = x - y with cost d */
    const int64_t N = (int64_t)n, M = (int64_t)m, target = N - M;
    std::vector<int64_t> cur(1, 0), nxt;
    auto slide = [&](int64_t x, int64_t k) {
        int64_t y = x - k;
        while (x < N && y < M && a[x] == b[y]) {
            ++x;
            ++y;
        }
        return x;
    };
    cur[0] = slide(0, 0);
    for (int64_t d = 0;; ++d) {
        if (target >= -d && target <= d && cur[(size_t)(target + d)] >= N) return (uint64_t)d;
        nxt.assign((size_t)(2 * d + 3), -1);
        for (int64_t k = -d - 1; k <= d + 1; ++k) {
            int64_t best = -1;
            if (k >= -d && k <= d && cur[(size_t)(k + d)] >= 0) best = cur[(size_t)(k + d)] + 1;                 /* substitution */
            if (k - 1 >= -d && k - 1 <= d && cur[(size_t)(k - 1 + d)] >= 0 && cur[(size_t)(k - 1 + d)] + 1 > best) best = cur[(size_t)(k - 1 + d)] + 1; /* a[x] unmatched */
            if (k + 1 >= -d && k + 1 <= d && cur[(size_t)(k + 1 + d)] > best) best = cur[(size_t)(k + 1 + d)]; /* b[y] unmatched */
            if (best < 0) continue;
            if (best > N) best = N;
            if (best - k > M) best = M + k; /* y must stay within b */
            if (best < 0 || best - k < 0) continue;
            nxt[(size_t)(k + d + 1)] = slide(best, k);
        }
        cur.swap(nxt);
    }
}

/* contig_base[c] = offset of contig c in the concatenated reference, contig_len[c] its length.
 * Two passes: a serial one that checks the variant ranges and lays out the per-variant outputs and the blobs
 * (prefix sums), then the regions are validated and their blobs written by `threads` workers (regions are
 * independent once the offsets are known). */
inline int pack_batch(const avk_region_batch *b, const std::vector<uint64_t> &contig_base, const std::vector<uint64_t> &contig_len,
                      const uint64_t *seq_off, const uint32_t *seq_stride, PackedBatch *out, std::string *err, int threads = 0, uint32_t lane_max_est = 15,
                      bool lane_pairs = true, uint32_t het_min = AVK_HET_SEARCH_MIN) {
    const uint64_t n = b->n_regions;
    if (n > 0x7FFFFFFFull || b->n_variants > 0x7FFFFFFFull) {
        *err = "batch too large (more than 2^31 regions or variants); split it";
        return AVK_E_ARG;
    }
    out->regions.resize(n);
    out->zyg_flags.assign(n, 0);
    out->nhet_u.assign(n, 0);
    out->delta_t.assign(n, 0);
    out->delta_q.assign(n, 0);
    out->fast_class.assign(n, 0);
    out->fast_key.assign(n, 0);
    /* the same cut of the regions into ranges for both passes */
    int nt = threads > 0 ? threads : (int)avk_usable_cpus();
    if (nt > 16) nt = 16;
    if ((uint64_t)nt > n / 4096 + 1) nt = (int)(n / 4096 + 1);
    if (nt < 1) nt = 1;
    auto range_lo = [&](int t) { return n * (uint64_t)t / (uint64_t)nt; };
    auto run_ranges = [&](const std::function<int(int, std::string *)> &fn) -> int {
        std::vector<int> rcs((size_t)nt, 0);
        std::vector<std::string> errs((size_t)nt);
        if (nt == 1) rcs[0] = fn(0, &errs[0]);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; ++t) pool.emplace_back([&, t] { rcs[(size_t)t] = fn(t, &errs[(size_t)t]); });
            for (auto &th : pool) th.join();
        }
        for (int t = 0; t < nt; ++t)
            if (rcs[(size_t)t]) {
                *err = errs[(size_t)t];
                return rcs[(size_t)t];
            }
        return 0;
    };
    /* pass 1: sizes — variant records and blob words per region (kept in the record), summed per range */
    std::vector<uint64_t> nv_at((size_t)nt + 1, 0), words_at((size_t)nt + 1, 0);
    int rc1 = run_ranges([&](int t, std::string *werr) -> int {
        uint64_t nv_t = 0, words_t = 0;
        for (uint64_t r = range_lo(t); r < range_lo(t + 1); ++r) {
            AvkDevRegion &dr = out->regions[r];
            memset(&dr, 0, sizeof(dr));
            const uint32_t tc = b->t_cnt[r], qc = b->q_cnt[r];
            uint64_t alle = 0;
            for (int side = 0; side < 2; ++side) {
                const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
                const uint32_t cnt = side == 0 ? tc : qc;
                if (off > b->n_variants || (uint64_t)cnt > b->n_variants - off) {
                    *werr = "variant range of a region exceeds n_variants";
                    return AVK_E_ARG;
                }
                for (uint32_t i = 0; i < cnt; ++i) alle += (uint64_t)b->a0_len[off + i] + b->a1_len[off + i];
            }
            const uint64_t N = (uint64_t)tc + qc;
            dr.t_cnt = tc;
            dr.q_cnt = qc;
            nv_t += N;
            uint64_t bytes = 0;
            if (N <= 60000) bytes = (((uint64_t)N * sizeof(AvkBlobVar) + 15) & ~15ull) + ((alle + 15) & ~15ull) + (uint64_t)N * sizeof(AvkOrdVar) + 32;
            if (bytes > 0x7FFFFFFFull) {
                *werr = "region blob exceeds 2 GiB; split the region's alleles";
                return AVK_E_ARG;
            }
            dr.blob_bytes = (uint32_t)bytes;
            dr.alle_bytes = (uint32_t)(alle < 0xFFFFFFFFull ? alle : 0xFFFFFFFFull);
            words_t += bytes / 4;
        }
        nv_at[(size_t)t + 1] = nv_t;
        words_at[(size_t)t + 1] = words_t;
        return 0;
    });
    if (rc1) return rc1;
    for (int t = 0; t < nt; ++t) {
        nv_at[(size_t)t + 1] += nv_at[(size_t)t];
        words_at[(size_t)t + 1] += words_at[(size_t)t];
    }
    const uint64_t nv = nv_at[(size_t)nt], n_words = words_at[(size_t)nt];
    if (nv > 0x7FFFFFFFull) {
        *err = "more than 2^31 variant records; split the batch";
        return AVK_E_ARG;
    }
    if (n_words / 2 > 0xFFFFFFFFull) {
        *err = "region blob arena exceeds its limits; split the batch";
        return AVK_E_ARG;
    }
    out->variants.resize(nv);
    out->dev2host.resize(nv);
    out->alleles.assign(1, 0);
    out->blob.resize(n_words ? n_words : 2);
    if (!n_words) out->blob[0] = out->blob[1] = 0;

    /* pass 2: validation and blobs, region by region */
    auto pack_range = [&](int t, std::string *werr) -> int {
        uint64_t v_at = nv_at[(size_t)t], w_at = words_at[(size_t)t]; /* running offsets of this range */
        for (uint64_t r = range_lo(t); r < range_lo(t + 1); ++r) {
            AvkDevRegion &dr = out->regions[r];
            const uint64_t blob_word = w_at;
            dr.v_off = (uint32_t)v_at;
            v_at += (uint64_t)dr.t_cnt + dr.q_cnt;
            w_at += dr.blob_bytes / 4;
            if (dr.blob_bytes) memset(out->blob.data() + blob_word, 0, dr.blob_bytes); /* padding between the parts stays zero */
            const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
            const uint64_t start = b->start[r], end = b->end[r];
            uint32_t pre = 0;
            if (c >= contig_len.size() || start > end || end > contig_len[c] || end - start > 0x7FFFFFFFull) pre = AVK_ST_INVALID_INPUT;
            const uint32_t tc = dr.t_cnt, qc = dr.q_cnt, N = tc + qc;
            if ((uint64_t)tc + qc > 60000) pre = AVK_ST_INVALID_INPUT;
            dr.len = pre ? 0 : (uint32_t)(end - start);
            dr.ref_off = pre ? 0 : contig_base[c] + start;
            AvkDevVariant *hv = out->variants.data() + dr.v_off;
            bool bad_zyg = false;
            uint64_t g[2] = {0, 0};
            uint32_t types = 0, counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t k = 0;
            for (int side = 0; side < 2; ++side) {
                const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
                const uint32_t cnt = side == 0 ? tc : qc;
                uint64_t last = 0;
                for (uint32_t i = 0; i < cnt; ++i, ++k) {
                    const uint64_t v = off + i;
                    AvkDevVariant &dv = hv[k];
                    const uint64_t pos = b->var_pos[v];
                    const uint32_t l0 = b->a0_len[v], l1 = b->a1_len[v];
                    const uint32_t raw = b->var_raw_space ? b->var_raw_space[v] : (l0 > l1 ? l0 : l1);
                    if (b->a0_off[v] + l0 > b->allele_bytes_len || b->a1_off[v] + l1 > b->allele_bytes_len) {
                        *werr = "allele range exceeds allele_bytes_len";
                        return AVK_E_ARG;
                    }
                    const uint8_t vt = b->var_type[v], zy = b->var_zyg[v];
                    if (l0 == 0 || l1 == 0 || raw < (l0 > l1 ? l0 : l1)) pre = AVK_ST_INVALID_INPUT;
                    if (vt >= AVK_N_VARIANT_TYPES || zy > AVK_ZYG_HOM_ALT) pre = AVK_ST_INVALID_INPUT;
                    if (pos < start || pos + l0 > end || pos < last) pre = AVK_ST_INVALID_INPUT;
                    last = pos;
                    if (zy == AVK_ZYG_UNKNOWN || zy == AVK_ZYG_HOM_REF) bad_zyg = true;
                    if (zy == AVK_ZYG_UNKNOWN) out->zyg_flags[r] |= 1;
                    if (zy == AVK_ZYG_HOM_REF) out->zyg_flags[r] |= 2;
                    if (zy == AVK_ZYG_UNPHASED_HET && out->nhet_u[r] < 255) out->nhet_u[r] += 1;
                    {
                        const int64_t w = zy == AVK_ZYG_HOM_ALT ? 2 : ((zy >= AVK_ZYG_UNPHASED_HET && zy <= AVK_ZYG_PHASED_HET10) ? 1 : 0);
                        (side == 0 ? out->delta_t[r] : out->delta_q[r]) += ((int64_t)l1 - (int64_t)l0) * w;
                    }
                    dv.rel_pos = pos >= start ? (uint32_t)(pos - start) : 0;
                    dv.a0_len = l0;
                    dv.a1_len = l1;
                    dv.a_off = 0;
                    dv.raw_space = raw;
                    dv.type = vt;
                    dv.zyg = zy;
                    out->dev2host[dr.v_off + k] = v;
                    if (l1 > l0) g[side] += l1 - l0;
                    if (vt < AVK_N_VARIANT_TYPES) {
                        types |= 1u << vt;
                        for (int t = 0; t < 8; ++t)
                            if (vt == AVK_SUP_TYPES[t]) counts[t] += side == 0 ? 1u : 0x10000u;
                    }
                }
            }
            if (!pre && bad_zyg) pre = AVK_ST_BAD_ZYGOSITY;
            if (!pre && (g[0] > g[1] ? g[0] : g[1]) > 0x7FFFFFFFull) pre = AVK_ST_INVALID_INPUT;
            dr.pre_status = pre;
            if (seq_off && seq_stride) {
                dr.seq_off = seq_off[r];
                dr.seq_stride = seq_stride[r];
            }
            if (pre) {
                dr.blob_bytes = 0;
                continue;
            }
            /* the region's blob */
            const uint64_t alle = dr.alle_bytes;
            const uint64_t vb = ((uint64_t)N * sizeof(AvkBlobVar) + 15) & ~15ull, ab = (alle + 15) & ~15ull, ob = (uint64_t)N * sizeof(AvkOrdVar);
            uint8_t *base = (uint8_t *)(out->blob.data() + blob_word);
            AvkBlobVar *bv = (AvkBlobVar *)base;
            uint8_t *ba = base + vb;
            AvkOrdVar *bo = (AvkOrdVar *)(base + vb + ab);
            uint32_t *bc = (uint32_t *)(base + vb + ab + ob);
            uint32_t run = 0;
            uint64_t ed_sum = 0;
            k = 0;
            for (int side = 0; side < 2; ++side) {
                const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
                const uint32_t cnt = side == 0 ? tc : qc;
                for (uint32_t i = 0; i < cnt; ++i, ++k) {
                    const uint64_t v = off + i;
                    const uint8_t *a0 = b->allele_bytes + b->a0_off[v], *a1 = b->allele_bytes + b->a1_off[v];
                    bv[k].rel_pos = hv[k].rel_pos;
                    bv[k].a0_len = hv[k].a0_len;
                    bv[k].a1_len = hv[k].a1_len;
                    bv[k].a_off = run;
                    bv[k].raw_space = hv[k].raw_space;
                    const uint64_t ed = host_edit_distance(a0, hv[k].a0_len, a1, hv[k].a1_len);
                    bv[k].alt_ed = (uint32_t)ed;
                    ed_sum += ed;
                    bv[k].type_zyg = (uint32_t)hv[k].type | ((uint32_t)hv[k].zyg << 8);
                    memcpy(ba + run, a0, hv[k].a0_len);
                    memcpy(ba + run + hv[k].a0_len, a1, hv[k].a1_len);
                    run += hv[k].a0_len + hv[k].a1_len;
                }
            }
            /* order_variants: stable by position over [truth.., query..]; each side is already sorted (checked above) */
            {
                uint32_t i = 0, j = tc, o = 0;
                while (i < tc || j < N) {
                    const uint32_t k2 = (j >= N || (i < tc && hv[i].rel_pos <= hv[j].rel_pos)) ? i++ : j++;
                    AvkOrdVar &ov = bo[o++];
                    ov.rel_pos = bv[k2].rel_pos;
                    ov.a0_len = bv[k2].a0_len;
                    ov.a1_len = bv[k2].a1_len;
                    ov.a_off = bv[k2].a_off;
                    ov.alt_ed = bv[k2].alt_ed;
                    ov.type_zyg = bv[k2].type_zyg;
                    ov.sync = dr.len;
                    ov.vi = k2;
                }
                for (uint32_t o2 = 0; o2 + 1 < N; ++o2) bo[o2].sync = bo[o2 + 1].rel_pos;
            }
            for (int t = 0; t < 8; ++t) bc[t] = counts[t];
            dr.blob_off = (uint32_t)(blob_word / 2);
            dr.grow = (uint32_t)(g[0] > g[1] ? g[0] : g[1]);
            dr.pre_status |= types << 16;
            dr.ed_bound = (uint32_t)(ed_sum < 0x7FFFFFFFull ? ed_sum : 0x7FFFFFFFull);
            /* small enough for the lane-per-region kernel?  (capacities of AvkFastClass; everything must fit the record's bit fields) */
            if (tc <= AVK_FAST_MAXV && qc <= AVK_FAST_MAXV && N >= 1 && dr.len <= 255 && ed_sum <= 255) {
                bool ok = true;
                for (uint32_t i = 0; i < N && ok; ++i) {
                    ok = bv[i].rel_pos <= 255 && bv[i].a0_len <= 255 && bv[i].a1_len <= 32 && bv[i].alt_ed <= 255 && bv[i].raw_space <= 0xFFFF;
                    const uint8_t *a1 = ba + bv[i].a_off + bv[i].a0_len;
                    for (uint32_t j = 0; j < bv[i].a1_len && ok; ++j) ok = a1[j] == 'A' || a1[j] == 'C' || a1[j] == 'G' || a1[j] == 'T';
                }
                for (int cl = 0; cl < AVK_FAST_GENERIC && ok; ++cl) {
                    const AvkFastClass &fc = AVK_FAST_CLASS[cl];
                    if (tc <= fc.maxv && qc <= fc.maxv && (uint64_t)dr.len + dr.grow <= 16ull * fc.W) {
                        out->fast_class[r] = (uint8_t)(cl + 1);
                        FastCall fcv[2 * AVK_FAST_MAXV];
                        for (uint32_t i = 0; i < N; ++i)
                            fcv[i] = FastCall{bv[i].rel_pos, bv[i].a0_len, bv[i].a1_len, bv[i].alt_ed, bv[i].type_zyg & 0xFFu, (bv[i].type_zyg >> 8) & 0xFFu, ba + bv[i].a_off + bv[i].a0_len};
                        out->fast_key[r] = fast_cost_key(fcv, tc, fcv + tc, qc);
                        if ((uint32_t)(out->fast_key[r] >> 4) > lane_max_est) out->fast_class[r] = 0; /* many edits: a whole wavefront's work (option lane_max_est) */
                        /* The three-call class keeps the caller's order: its expensive regions are large searches, which the key does not
                         * see, and lanes that diverge do not run side by side — 16 expensive regions in one tile take 16 times as long as
                         * one, and the launch lasts as long as that tile (measured: 3.1 ms in caller order, 6.2 ms sorted). */
                        if (fc.maxv > 2) out->fast_key[r] = 0;
                        if (fc.maxv > 2 && het_min && out->nhet_u[r] >= het_min) out->fast_class[r] = 0; /* a big phasing search (avk_dev_types.h): not for a lane */
                        break;
                    }
                }
                /* the same SNV on both sides: looked up, not searched (avk_pairs.inl) */
                if (ok && out->fast_class[r] && lane_pairs && tc == 1 && qc == 1) {
                    auto code = [&](const AvkBlobVar &x) { const uint8_t ch = ba[x.a_off + x.a0_len]; return ch == 'C' ? 1u : (ch == 'G' ? 2u : (ch == 'T' ? 3u : 0u)); };
                    if (pairs::pair_is_candidate(tc, qc, bv[0].rel_pos, bv[1].rel_pos, bv[0].a0_len, bv[0].a1_len, bv[1].a0_len, bv[1].a1_len, bv[0].type_zyg & 0xFFu, bv[1].type_zyg & 0xFFu,
                                                 (bv[0].type_zyg >> 8) & 0xFFu, (bv[1].type_zyg >> 8) & 0xFFu, bv[0].alt_ed, bv[1].alt_ed, bv[0].raw_space, bv[1].raw_space,
                                                 bv[0].a1_len ? code(bv[0]) : 0u, bv[1].a1_len ? code(bv[1]) : 0u)) {
                        out->fast_class[r] = (uint8_t)(AVK_FAST_PAIR + 1);
                        out->fast_key[r] = 0;
                    }
                }
            }
        }
        return 0;
    };
    return run_ranges(pack_range);
}


/* Work order of the first launch and the prediction of which regions outgrow the LDS slices.
 * order = [class C | class B | the rest], each part with the most variants first:
 *   class C  predicted to outgrow even a tier-1 slice (upper range, about 6 N nodes alive = class_c_nodes_x2 / 2): solved by the HBM solo launch, which shares its list with the main stream's HBM launch;
 *   class B  predicted to outgrow the small slice (typical case, about 2N+1 nodes alive) or with at least
 *            solo_min_variants variants: solved by the solo waves with a tier-1 slice each;
 *   the rest goes to the bulk launch.
 * The prediction mirrors the workspace carve of solve_region_tier (avk_solver.inl).  A wrong guess only costs
 * time: a region solved in a larger class had more room than it needed, a missed one overflows into the next
 * tier's launch as before. */
/* bytes of a wave's slice in the bulk launch: the workgroup's LDS (4 x lds_bytes_per_wave) minus its tail, in 4 equal parts */
inline uint64_t bulk_slice_bytes(uint64_t lds_bytes_per_wave) {
    if (lds_bytes_per_wave < 1024) return lds_bytes_per_wave;
    return ((4 * lds_bytes_per_wave - AVK_WG_TAIL_BYTES) / 4) & ~15ull;
}

/* the region records in work order, each remembering where it came from */
inline PodVec<AvkDevRegion> regions_in_work_order(const PackedBatch &pb, const std::vector<uint32_t> &order) {
    PodVec<AvkDevRegion> out;
    out.resize(order.size());
    const size_t n = order.size();
    size_t nt = avk_usable_cpus();
    if (nt > 16) nt = 16;
    if (nt > n / 65536 + 1) nt = n / 65536 + 1;
    if (nt < 1) nt = 1;
    auto part = [&](size_t t) {
        for (size_t k = n * t / nt; k < n * (t + 1) / nt; ++k) {
            out[k] = pb.regions[order[k]];
            out[k].orig = order[k];
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(part, t);
    part(0);
    for (auto &th : pool) th.join();
    return out;
}

struct WorkPlan {
    uint32_t n_hbm = 0;  /* class C */
    uint32_t n_hbm_notwide = 0; /* of those (by size): not for avk_wide.inl by their record (avk_wide_static_ok) */
    uint32_t n_hard = 0; /* class B */
    /* the regions of the lane-per-region kernel's classes close the work order, largest class first:
     * [class C | class B | bulk | fast class AVK_FAST_CLASSES - 1 | .. | fast class 0] */
    uint32_t n_fast[AVK_FAST_CLASSES] = {0};
    uint32_t n_fast_heavy[AVK_FAST_CLASSES] = {0}; /* of those, regions with estimated edits (fast_cost_key >> 4 != 0): they lead the class's records */
    uint32_t fast_base[AVK_FAST_CLASSES] = {0}; /* first record of the class in the work order */
    uint32_t n_fast_total = 0;
};

/* lane_min_regions: a class of the lane-per-region kernel is used when the batch holds at least lane_min_regions x {1, 1, 16, 16} of its
 * regions (a launch lasts as long as its slowest tile: a small class is better off with the wave-per-region kernels); the regions of
 * an unused class are planned like any other region.  0xFFFFFFFF = no lane classes at all. */
inline WorkPlan plan_work_order(const PackedBatch &pb, uint64_t tier0_bytes, uint32_t tier0_ed_cap, uint64_t tier1_bytes, uint32_t tier1_ed_cap,
                                uint32_t solo_min_variants, uint32_t max_branch, std::vector<uint32_t> *order, uint32_t class_c_nodes_x2 = 12,
                                uint64_t lane_min_regions = 0, uint32_t lane_max_calls = AVK_FAST_MAXV, uint64_t lane_min_batch = 0, uint32_t stripe_w = 0,
                                uint32_t head_est = 1, uint32_t het_min = AVK_HET_SEARCH_MIN, uint64_t class_c_below = 0) {
    const uint64_t n = pb.regions.size();
    order->assign(n, 0);
    std::vector<uint8_t> cls(n, 2); /* 0 = C, 1 = B, 2 = bulk, 3 + k = fast class AVK_FAST_CLASSES - 1 - k */
    WorkPlan plan;
    bool lane_on[AVK_FAST_CLASSES];
    {
        static const uint64_t scale[AVK_FAST_CLASSES] = {1, 1, 16, 16, 2, 1};
        uint64_t have[AVK_FAST_CLASSES] = {0};
        if (!pb.fast_class.empty())
            for (uint64_t r = 0; r < n; ++r)
                if (pb.fast_class[r] && !(pb.regions[r].pre_status & 0xFFFFu)) have[pb.fast_class[r] - 1u] += 1;
        for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc)
            lane_on[fc] = lane_min_regions != 0xFFFFFFFFull && have[fc] > 0 && have[fc] >= lane_min_regions * scale[fc] && AVK_FAST_CLASS[fc].maxv <= lane_max_calls;
        /* a resident batch that is small altogether is better off without the lane launches: a step is then one chain of five launches
         * instead of a dozen on six streams, and a lane launch cannot be shorter than one tile (chr20, 48 k regions: 0.37 ms per
         * synchronised step without, 0.41 with; queued back to back 0.40 / 0.65) */
        uint64_t have_all = 0;
        for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) have_all += lane_on[fc] ? have[fc] : 0;
        if (lane_min_regions != 0 && have_all < lane_min_batch)
            for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) lane_on[fc] = false;
    }
    bool lanes_any = false;
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) lanes_any = lanes_any || lane_on[fc];
    /* class_c_below: a batch with lane launches and few regions outside them (a contig, the shard of a rank) plans those the wide kernel can take as class C — what
     * is left of such a step once the lanes are done is the latency of single regions, and the wave-cooperative kernel has the shortest (profiles/r05_small_batches.txt) */
    bool few_outside = false;
    if (lanes_any && class_c_below) {
        uint64_t in_lanes = 0;
        for (uint64_t r = 0; r < n; ++r)
            in_lanes += !pb.fast_class.empty() && pb.fast_class[r] && !(pb.regions[r].pre_status & 0xFFFFu) && lane_on[pb.fast_class[r] - 1u];
        few_outside = n - in_lanes <= class_c_below;
    }
    auto need = [&](const AvkDevRegion &dr, uint64_t N, uint64_t alle, uint64_t grow, uint32_t tier_cap, uint64_t nodes) {
        const uint64_t seqcap = ((uint64_t)dr.len + grow + 7) & ~7ull;
        const uint64_t maxT = dr.t_cnt > dr.q_cnt ? dr.t_cnt : dr.q_cnt;
        const uint64_t alw = maxT ? (maxT + 63) >> 6 : 1;
        uint64_t cap = tier_cap;
        if (cap && dr.ed_bound < cap) cap = dr.ed_bound ? dr.ed_bound : 1;
        uint64_t wfcap = cap ? 2 * cap + 2 : 2 * seqcap + 4;
        if (wfcap > 2 * seqcap + 4) wfcap = 2 * seqcap + 4;
        const uint64_t hapA = (48 + 16 * alw + 4 * wfcap + 2 * seqcap + 15) & ~15ull, nodeA = 16 + 2 * hapA;
        const uint64_t optcap = max_branch < 4096 ? max_branch : 4096;
        const uint64_t fixed = dr.len + 8 + 28 * N + alle + 32 + 32 * N + 4 * N + 16 + 32 + 4 * optcap + 8 * 8 * alw + 8 * 4 * alw * 8 + 32 + 64;
        return fixed + nodes * (nodeA + 16);
    };
    for (uint64_t r = 0; r < n; ++r) {
        const AvkDevRegion &dr = pb.regions[r];
        const uint64_t N = (uint64_t)dr.t_cnt + dr.q_cnt;
        if (!pb.fast_class.empty() && pb.fast_class[r] && !(dr.pre_status & 0xFFFFu) && lane_on[pb.fast_class[r] - 1u]) {
            const uint32_t fc = pb.fast_class[r] - 1u;
            cls[r] = (uint8_t)(3 + (AVK_FAST_CLASSES - 1 - fc));
            plan.n_fast[fc] += 1;
            plan.n_fast_heavy[fc] += (uint32_t)(pb.fast_key[r] >> 4) >= (head_est ? head_est : 1u);
            continue;
        }
        if ((dr.pre_status & 0xFFFFu) || N == 0 || solo_min_variants == 0) continue;
        const uint64_t alle = dr.alle_bytes, grow = dr.grow;
        if (tier1_bytes && ((lanes_any && het_min && !pb.nhet_u.empty() && pb.nhet_u[r] >= het_min) || /* (a big phasing search, in a batch with lane launches) */ need(dr, N, alle, grow, tier1_ed_cap, ((uint64_t)class_c_nodes_x2 * N + 1) / 2) > tier1_bytes)) {
            cls[r] = 0;
            plan.n_hbm += 1;
            if (need(dr, N, alle, grow, tier1_ed_cap, ((uint64_t)class_c_nodes_x2 * N + 1) / 2) > tier1_bytes && !avk_wide_static_ok(dr.len, dr.grow, dr.ed_bound, dr.t_cnt, dr.q_cnt, 0u))
                plan.n_hbm_notwide += 1;
        } else if (few_outside && tier1_bytes && avk_wide_static_ok(dr.len, dr.grow, dr.ed_bound, dr.t_cnt, dr.q_cnt, 0u)) {
            cls[r] = 0;
            plan.n_hbm += 1;
        } else if (N >= solo_min_variants || need(dr, N, alle, grow, tier0_ed_cap, 2 * N + 1) > tier0_bytes) {
            cls[r] = 1;
            plan.n_hard += 1;
        }
    }
    /* counting sort by class, then by variant count (lane classes: by fast_cost_key), descending */
    std::vector<uint64_t> cnt((3 + AVK_FAST_CLASSES) * 256 + 1, 0);
    auto key = [&](uint64_t r) {
        uint32_t k = pb.regions[r].t_cnt + pb.regions[r].q_cnt;
        k = k > 255u ? 255u : k;
        if (cls[r] >= 3) k = pb.fast_key[r]; /* a tile of 64 lanes should hold regions of one cost */
        return 256u * cls[r] + (255u - k);
    };
    for (uint64_t r = 0; r < n; ++r) cnt[key(r) + 1] += 1;
    for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
    for (uint64_t r = 0; r < n; ++r) (*order)[cnt[key(r)]++] = (uint32_t)r;
    {
        uint32_t at = (uint32_t)n;
        for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) { /* class 0 is the last segment */
            at -= plan.n_fast[fc];
            plan.fast_base[fc] = at;
            plan.n_fast_total += plan.n_fast[fc];
        }
    }
    /* the heads of the lane classes are dealt out over their claims (avk_stripe_slot, avk_dev_types.h) */
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) {
        const uint32_t hs = avk_head_slots(AVK_FAST_CLASS[fc].maxv, plan.n_fast[fc], plan.n_fast_heavy[fc], stripe_w);
        if (!hs) continue;
        std::vector<uint32_t> head(order->begin() + plan.fast_base[fc], order->begin() + plan.fast_base[fc] + hs);
        for (uint32_t p = 0; p < hs; ++p) (*order)[plan.fast_base[fc] + avk_stripe_slot(p, hs, stripe_w)] = head[p];
    }
    return plan;
}

/* 2 bits per base, 16 bases per word (A 0, C 1, G 2, T 3), the layout of the packed reference */
inline uint32_t pack_bases_2bit(const uint8_t *s, uint32_t n) {
    uint32_t w = 0;
    for (uint32_t i = 0; i < n && i < 16; ++i) {
        const uint8_t ch = s[i];
        const uint32_t code = ch == 'C' ? 1u : (ch == 'G' ? 2u : (ch == 'T' ? 3u : 0u));
        w |= code << (2 * i);
    }
    return w;
}

/* The fast records (avk_dev_types.h) of the plan's fast segments, class AVK_FAST_CLASSES - 1 first (the order of the segments).
 * word_base[c] = first word of class c's tiles in the returned array, n_tiles[c] = its tiles (AVK_FAST_WORDS_OF(maxv) * 64 words each). */
inline PodVec<uint32_t> build_fast_records(const PackedBatch &pb, const std::vector<uint32_t> &order, const WorkPlan &plan, uint64_t *word_base,
                                           uint32_t *n_tiles) {
    uint64_t words = 0;
    uint32_t tiles = 0, tile_first[AVK_FAST_CLASSES];
    for (int fc = AVK_FAST_CLASSES - 1; fc >= 0; --fc) {
        word_base[fc] = words;
        tile_first[fc] = tiles;
        n_tiles[fc] = (plan.n_fast[fc] + 63u) / 64u;
        tiles += n_tiles[fc];
        words += (uint64_t)n_tiles[fc] * AVK_FAST_WORDS_OF(AVK_FAST_CLASS[fc].maxv) * 64u;
    }
    PodVec<uint32_t> recs;
    recs.resize((size_t)words + 1);
    size_t nt = avk_usable_cpus();
    if (nt > 16) nt = 16;
    if (nt > tiles / 256 + 1) nt = tiles / 256 + 1;
    if (nt < 1) nt = 1;
    auto part = [&](size_t t) {
        for (uint32_t tile = (uint32_t)((uint64_t)tiles * t / nt); tile < (uint32_t)((uint64_t)tiles * (t + 1) / nt); ++tile) {
            int fc = 0;
            for (int c = 0; c < AVK_FAST_CLASSES; ++c)
                if (tile >= tile_first[c] && tile < tile_first[c] + n_tiles[c]) fc = c;
            const uint32_t maxv = AVK_FAST_CLASS[fc].maxv, rw = AVK_FAST_WORDS_OF(maxv);
            uint32_t *T = recs.data() + word_base[fc] + (size_t)(tile - tile_first[fc]) * rw * 64u;
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t k = (tile - tile_first[fc]) * 64u + lane;
                for (uint32_t w = 0; w < rw; ++w) T[w * 64 + lane] = w == 1 ? 0xFFFFFFFFu : 0u;
                if (k >= plan.n_fast[fc]) continue;
                const uint32_t r = order[plan.fast_base[fc] + k];
                const AvkDevRegion &dr = pb.regions[r];
                const uint32_t tc = dr.t_cnt, qc = dr.q_cnt, N = tc + qc;
                const uint8_t *base = (const uint8_t *)(pb.blob.data() + 2ull * dr.blob_off);
                const AvkBlobVar *bv = (const AvkBlobVar *)base;
                const uint8_t *ba = base + (((uint64_t)N * sizeof(AvkBlobVar) + 15) & ~15ull);
                const AvkOrdVar *bo = (const AvkOrdVar *)(ba + (((uint64_t)dr.alle_bytes + 15) & ~15ull));
                uint32_t ord = 0;
                for (uint32_t d = 0; d < N; ++d) {
                    const uint32_t vi = bo[d].vi;
                    ord |= (vi < tc ? 0u : 1u) << d; /* each side's calls come in their own order */
                }
                T[0 * 64 + lane] = (uint32_t)(dr.ref_off >> 4);
                T[1 * 64 + lane] = (uint32_t)(dr.ref_off & 15u) | (dr.len << 4) | (tc << 12) | (qc << 14) | (ord << 16);
                T[2 * 64 + lane] = dr.v_off;
                T[3 * 64 + lane] = r;
                for (uint32_t i = 0; i < N; ++i) {
                    const uint32_t slot = i < tc ? i : maxv + (i - tc);
                    uint32_t *V = T + (AVK_FAST_HDR + 4 * slot) * 64 + lane;
                    const uint8_t *a1 = ba + bv[i].a_off + bv[i].a0_len;
                    V[0] = bv[i].rel_pos | (bv[i].a0_len << 8) | (bv[i].a1_len << 16) | ((bv[i].type_zyg & 0xFu) << 24) | (((bv[i].type_zyg >> 8) & 7u) << 28);
                    V[64] = bv[i].alt_ed | (bv[i].raw_space << 8);
                    V[128] = pack_bases_2bit(a1, bv[i].a1_len);
                    V[192] = bv[i].a1_len > 16 ? pack_bases_2bit(a1 + 16, bv[i].a1_len - 16) : 0u;
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(part, t);
    part(0);
    for (auto &th : pool) th.join();
    return recs;
}

} // namespace avk
#endif
