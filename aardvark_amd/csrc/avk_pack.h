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

#include <cstring>
#include <string>
#include <vector>

#include "avk_dev_types.h"

namespace avk {

struct PackedBatch {
    std::vector<AvkDevRegion> regions;
    std::vector<AvkDevVariant> variants;
    std::vector<uint8_t> alleles;
    std::vector<uint64_t> dev2host; /* device variant index -> caller variant index */
    std::vector<uint8_t> zyg_flags; /* per region: bit 0 an Unknown zygosity, bit 1 a HomozygousReference one */
    std::vector<int64_t> delta_t, delta_q; /* variant_delta_length per side (merge_solver.rs:211-223) */
    uint64_t seq_total = 0;
};

inline uint32_t seq_stride_of(const avk_region_batch *b, uint64_t r) {
    uint64_t L = b->end[r] >= b->start[r] ? b->end[r] - b->start[r] : 0;
    uint64_t g[2] = {0, 0};
    for (int side = 0; side < 2; ++side) {
        uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
        uint32_t cnt = side == 0 ? b->t_cnt[r] : b->q_cnt[r];
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

/* contig_base[c] = offset of contig c in the concatenated reference, contig_len[c] its length */
inline int pack_batch(const avk_region_batch *b, const std::vector<uint64_t> &contig_base, const std::vector<uint64_t> &contig_len,
                      const uint64_t *seq_off, const uint32_t *seq_stride, PackedBatch *out, std::string *err) {
    const uint64_t n = b->n_regions;
    if (n > 0x7FFFFFFFull || b->n_variants > 0x7FFFFFFFull) {
        *err = "batch too large (more than 2^31 regions or variants); split it";
        return AVK_E_ARG;
    }
    out->regions.resize(n);
    out->zyg_flags.assign(n, 0);
    out->delta_t.assign(n, 0);
    out->delta_q.assign(n, 0);
    out->variants.clear();
    out->variants.reserve(b->n_variants);
    out->alleles.clear();
    out->dev2host.clear();
    out->dev2host.reserve(b->n_variants);
    for (uint64_t r = 0; r < n; ++r) {
        AvkDevRegion &dr = out->regions[r];
        memset(&dr, 0, sizeof(dr));
        const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
        const uint64_t start = b->start[r], end = b->end[r];
        uint32_t pre = 0;
        if (c >= contig_len.size() || start > end || end > contig_len[c] || end - start > 0x7FFFFFFFull) pre = AVK_ST_INVALID_INPUT;
        const uint32_t tc = b->t_cnt[r], qc = b->q_cnt[r];
        if ((uint64_t)tc + qc > 60000) pre = AVK_ST_INVALID_INPUT;
        dr.v_off = (uint32_t)out->variants.size();
        dr.t_cnt = tc;
        dr.q_cnt = qc;
        dr.len = pre ? 0 : (uint32_t)(end - start);
        dr.ref_off = pre ? 0 : contig_base[c] + start;
        bool bad_zyg = false;
        for (int side = 0; side < 2; ++side) {
            const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
            const uint32_t cnt = side == 0 ? tc : qc;
            uint64_t last = 0;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t v = off + i;
                if (v >= b->n_variants) {
                    *err = "variant range of a region exceeds n_variants";
                    return AVK_E_ARG;
                }
                AvkDevVariant dv;
                memset(&dv, 0, sizeof(dv));
                const uint64_t pos = b->var_pos[v];
                const uint32_t l0 = b->a0_len[v], l1 = b->a1_len[v];
                const uint32_t raw = b->var_raw_space ? b->var_raw_space[v] : (l0 > l1 ? l0 : l1);
                if (b->a0_off[v] + l0 > b->allele_bytes_len || b->a1_off[v] + l1 > b->allele_bytes_len) {
                    *err = "allele range exceeds allele_bytes_len";
                    return AVK_E_ARG;
                }
                if (l0 == 0 || l1 == 0 || raw < (l0 > l1 ? l0 : l1)) pre = AVK_ST_INVALID_INPUT;
                if (b->var_type[v] >= AVK_N_VARIANT_TYPES || b->var_zyg[v] > AVK_ZYG_HOM_ALT) pre = AVK_ST_INVALID_INPUT;
                if (pos < start || pos + l0 > end || pos < last) pre = AVK_ST_INVALID_INPUT;
                last = pos;
                if (b->var_zyg[v] == AVK_ZYG_UNKNOWN || b->var_zyg[v] == AVK_ZYG_HOM_REF) bad_zyg = true;
                if (b->var_zyg[v] == AVK_ZYG_UNKNOWN) out->zyg_flags[r] |= 1;
                if (b->var_zyg[v] == AVK_ZYG_HOM_REF) out->zyg_flags[r] |= 2;
                {
                    const int64_t w = b->var_zyg[v] == AVK_ZYG_HOM_ALT ? 2 : ((b->var_zyg[v] >= AVK_ZYG_UNPHASED_HET && b->var_zyg[v] <= AVK_ZYG_PHASED_HET10) ? 1 : 0);
                    (side == 0 ? out->delta_t[r] : out->delta_q[r]) += ((int64_t)l1 - (int64_t)l0) * w;
                }
                dv.rel_pos = pos >= start ? (uint32_t)(pos - start) : 0;
                dv.a0_len = l0;
                dv.a1_len = l1;
                dv.a_off = (uint32_t)out->alleles.size();
                dv.raw_space = raw;
                dv.type = b->var_type[v];
                dv.zyg = b->var_zyg[v];
                if (out->alleles.size() + (uint64_t)l0 + l1 > 0xFFFFFFF0ull) {
                    *err = "allele arena exceeds 4 GiB; split the batch";
                    return AVK_E_ARG;
                }
                out->alleles.insert(out->alleles.end(), b->allele_bytes + b->a0_off[v], b->allele_bytes + b->a0_off[v] + l0);
                out->alleles.insert(out->alleles.end(), b->allele_bytes + b->a1_off[v], b->allele_bytes + b->a1_off[v] + l1);
                out->variants.push_back(dv);
                out->dev2host.push_back(v);
            }
        }
        if (!pre && bad_zyg) pre = AVK_ST_BAD_ZYGOSITY;
        dr.pre_status = pre;
        if (seq_off && seq_stride) {
            dr.seq_off = seq_off[r];
            dr.seq_stride = seq_stride[r];
        }
    }
    if (out->alleles.empty()) out->alleles.push_back(0);
    return 0;
}

} // namespace avk
#endif
