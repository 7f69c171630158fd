/*
 * oracle.cpp — CPU ORACLE for the compare hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * A sequential C++ restatement of the reference algorithm, one function per reference
 * function, each citing the file:line of PacificBiosciences/aardvark v0.10.5 it follows.
 * Data structures are the plain ones the reference uses (growable byte vectors per haplotype,
 * one heap-allocated node per search state, a binary heap keyed by the same priority tuple);
 * nothing here is tuned.  Parity is pinned by tests/test_oracle_golden.py against the
 * reference's own known-answer tests (tests/golden/).
 */
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

typedef std::vector<uint8_t> Bytes;
struct Span {
    const uint8_t *p;
    size_t n;
    Span() : p(nullptr), n(0) {}
    Span(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    Span(const Bytes &b) : p(b.data()), n(b.size()) {}
};

enum { A_UNKNOWN = 0, A_REF = 1, A_ALT = 2 }; /* Allele, phase_enums.rs:20-24 */

/* statistics (sizing aid for the kernels) */
struct Stats {
    uint64_t max_pops_a = 0, max_queue_a = 0, max_pops_b = 0, max_queue_b = 0, max_ed = 0, max_optima = 0;
    uint64_t total_pops_a = 0, total_pops_b = 0, total_wfa = 0, incr_mismatch = 0;
    uint64_t byte_compares = 0; /* base comparisons made by DWFALite::extend (the compute-side figure of SURVEY.md 8d) */
    void merge(const Stats &o) {
        max_pops_a = std::max(max_pops_a, o.max_pops_a);
        max_queue_a = std::max(max_queue_a, o.max_queue_a);
        max_pops_b = std::max(max_pops_b, o.max_pops_b);
        max_queue_b = std::max(max_queue_b, o.max_queue_b);
        max_ed = std::max(max_ed, o.max_ed);
        max_optima = std::max(max_optima, o.max_optima);
        total_pops_a += o.total_pops_a;
        total_pops_b += o.total_pops_b;
        total_wfa += o.total_wfa;
        incr_mismatch += o.incr_mismatch;
        byte_compares += o.byte_compares;
    }
};
thread_local Stats t_stats;
Stats g_stats;
std::mutex g_stats_mutex;

/* ------------------------------------------------------------------------------------------
 * DWFALite — src/dwfa/dynamic_wfa.rs:23-276
 * ---------------------------------------------------------------------------------------- */
enum DErr { D_OK = 0, D_MAXED = 1, D_FINAL = 2 };

struct DWFALite {
    size_t edit_distance = 0;
    std::vector<size_t> wavefront{0}; /* length 2*ed+1, value = symbols of `other` consumed (:45,:152) */
    bool is_finalized = false;
    size_t max_edit_distance = SIZE_MAX;

    bool operator==(const DWFALite &o) const {
        return edit_distance == o.edit_distance && wavefront == o.wavefront && is_finalized == o.is_finalized &&
               max_edit_distance == o.max_edit_distance;
    }

    /* :94-130 */
    void extend(Span b, Span o) {
        uint64_t compares = 0;
        for (size_t i = 0; i < wavefront.size(); ++i) {
            size_t &d = wavefront[i];
            for (;;) {
                size_t baseline_offset = d + edit_distance - i;
                size_t other_offset = d;
                if (baseline_offset >= b.n || other_offset >= o.n) break;
                compares += 1;
                if (b.p[baseline_offset] != o.p[other_offset]) break;
                d += 1;
            }
        }
        t_stats.byte_compares += compares;
    }

    /* :140-173 — the distance is incremented BEFORE the max check (:146-149); offsets are not
     * clipped to the sequence lengths (:156-165) */
    DErr increase_edit_distance(Span b, Span o) {
        if (is_finalized) return D_FINAL;
        edit_distance += 1;
        if (edit_distance > max_edit_distance) return D_MAXED;
        t_stats.max_ed = std::max<uint64_t>(t_stats.max_ed, edit_distance);
        std::vector<size_t> nw(wavefront.size() + 2, 0);
        for (size_t i = 0; i < wavefront.size(); ++i) {
            size_t d = wavefront[i];
            nw[i] = std::max(nw[i], d);
            nw[i + 1] = std::max(nw[i + 1], d + 1);
            nw[i + 2] = std::max(nw[i + 2], d + 1);
        }
        wavefront.swap(nw);
        extend(b, o);
        return D_OK;
    }

    /* :201-215 */
    size_t maximum_baseline_distance() const {
        size_t m = 0;
        for (size_t i = 0; i < wavefront.size(); ++i) m = std::max(m, wavefront[i] + edit_distance - i);
        return m;
    }
    size_t maximum_other_distance() const { return *std::max_element(wavefront.begin(), wavefront.end()); }
    /* :220-245 */
    bool reached_baseline_end(Span b) const { return maximum_baseline_distance() >= b.n; }
    bool reached_other_end(Span o) const { return maximum_other_distance() >= o.n; }
    bool reached_full_diagonal(Span b, Span o) const {
        for (size_t i = 0; i < wavefront.size(); ++i) {
            size_t d = wavefront[i];
            if (d + edit_distance - i >= b.n && d >= o.n) return true;
        }
        return false;
    }

    /* :68-84 — stops as soon as EITHER end is touched */
    DErr update(Span b, Span o) {
        if (is_finalized) return D_FINAL;
        extend(b, o);
        while (!reached_baseline_end(b) && !reached_other_end(o)) {
            DErr e = increase_edit_distance(b, o);
            if (e) return e;
        }
        return D_OK;
    }

    /* :183-198 */
    DErr finalize(Span b, Span o) {
        if (is_finalized) return D_FINAL;
        extend(b, o);
        while (!reached_full_diagonal(b, o)) {
            DErr e = increase_edit_distance(b, o);
            if (e) return e;
        }
        is_finalized = true;
        return D_OK;
    }
};

/* util/sequence_alignment.rs:9-13 */
size_t wfa_ed(Span a, Span b) {
    t_stats.total_wfa += 1;
    DWFALite d;
    d.finalize(a, b); /* max ED unbounded: cannot fail */
    return d.edit_distance;
}

/* util/sequence_alignment.rs:20-51 — full two-row DP, v1 on the x axis */
size_t edit_distance(Span v1, Span v2) {
    size_t l1 = v1.n;
    std::vector<size_t> row(l1 + 1, 0), prev(l1 + 1);
    for (size_t j = 0; j <= l1; ++j) prev[j] = j;
    for (size_t i = 0; i < v2.n; ++i) {
        uint8_t c2 = v2.p[i];
        row[0] = i + 1;
        for (size_t j = 0; j < l1; ++j) {
            size_t a = prev[j + 1] + 1, b = row[j] + 1, c = prev[j] + (v1.p[j] == c2 ? 0 : 1);
            row[j + 1] = std::min(a, std::min(b, c));
        }
        row.swap(prev);
    }
    return prev[l1];
}

/* ------------------------------------------------------------------------------------------
 * Variant view — src/data_types/variants.rs:73-91
 * ---------------------------------------------------------------------------------------- */
struct Var {
    uint64_t position;
    Span allele0, allele1;
    uint8_t variant_type;
    uint8_t zyg; /* input PhasedZygosity */
    uint64_t raw_allele_space;
    size_t ref_len() const { return allele0.n; }              /* :430-432 */
    size_t alt_ed() const { return wfa_ed(allele0, allele1); } /* :413-415 */
};

/* ------------------------------------------------------------------------------------------
 * HaplotypeTracker / HaplotypeDWFA — src/dwfa/haplotype_dwfa.rs:17-245
 * ---------------------------------------------------------------------------------------- */
enum HErr { H_OK = 0, H_D_MAXED = 1, H_D_FINAL = 2, H_UNKNOWN_ALLELE = 3 };

struct HaplotypeTracker {
    size_t ref_pos;
    std::vector<uint8_t> alleles;
    Bytes sequence;
    size_t variant_skip_distance = 0;
    explicit HaplotypeTracker(size_t region_start) : ref_pos(region_start) {}

    /* :218-227 */
    void copy_reference(Span reference, size_t region_end) {
        if (ref_pos < region_end) {
            size_t e = std::min(region_end, reference.n); /* the reference panics past the contig end; inputs are validated */
            if (ref_pos < e) sequence.insert(sequence.end(), reference.p + ref_pos, reference.p + e);
            ref_pos = region_end;
        }
    }

    /* :175-212 */
    HErr extend_variant(Span reference, const Var &variant, int allele, bool has_ext, size_t ref_extension, bool *success) {
        size_t variant_start = (size_t)variant.position;
        copy_reference(reference, variant_start);
        bool ok;
        if (allele == A_UNKNOWN) return H_UNKNOWN_ALLELE;
        if (allele == A_REF) {
            ok = true; /* nothing appended, ref_pos unchanged (:183-186) */
        } else {
            if (ref_pos <= variant_start) {
                sequence.insert(sequence.end(), variant.allele1.p, variant.allele1.p + variant.allele1.n);
                ref_pos = variant_start + variant.ref_len();
                ok = true;
            } else {
                variant_skip_distance += edit_distance(variant.allele0, variant.allele1); /* :199 */
                ok = false;
            }
        }
        alleles.push_back((uint8_t)allele);
        if (has_ext) copy_reference(reference, ref_extension);
        *success = ok;
        return H_OK;
    }
};

struct HaplotypeDWFA {
    HaplotypeTracker truth_haplotype, query_haplotype;
    DWFALite dwfa;
    HaplotypeDWFA(size_t region_start, size_t max_ed) : truth_haplotype(region_start), query_haplotype(region_start) {
        dwfa.max_edit_distance = max_ed;
    }

    /* :72-78 */
    HErr update_dwfa() {
        DErr e = dwfa.update(truth_haplotype.sequence, query_haplotype.sequence);
        return e == D_OK ? H_OK : (e == D_MAXED ? H_D_MAXED : H_D_FINAL);
    }

    /* :46-67 */
    HErr extend_variant(Span reference, bool is_truth, const Var &variant, int allele, bool has_sync, size_t sync, bool *success) {
        HErr e;
        if (is_truth) {
            if (has_sync) query_haplotype.copy_reference(reference, sync);
            e = truth_haplotype.extend_variant(reference, variant, allele, has_sync, sync, success);
        } else {
            if (has_sync) truth_haplotype.copy_reference(reference, sync);
            e = query_haplotype.extend_variant(reference, variant, allele, has_sync, sync, success);
        }
        if (e) return e;
        return update_dwfa();
    }

    /* :84-95 */
    HErr finalize_dwfa(Span reference, size_t region_end) {
        truth_haplotype.copy_reference(reference, region_end);
        query_haplotype.copy_reference(reference, region_end);
        HErr e = update_dwfa();
        if (e) return e;
        DErr d = dwfa.finalize(truth_haplotype.sequence, query_haplotype.sequence);
        return d == D_OK ? H_OK : (d == D_MAXED ? H_D_MAXED : H_D_FINAL);
    }

    /* :99-112 */
    bool is_synchronized() const {
        return dwfa.edit_distance == 0 && truth_haplotype.sequence.size() == query_haplotype.sequence.size() &&
               truth_haplotype.ref_pos == query_haplotype.ref_pos;
    }
    size_t edit_distance() const { return dwfa.edit_distance; }
    size_t set_alleles() const { return truth_haplotype.alleles.size() + query_haplotype.alleles.size(); } /* :120 */
    size_t total_skip() const { return truth_haplotype.variant_skip_distance + query_haplotype.variant_skip_distance; }
    size_t total_cost() const { return edit_distance() + total_skip(); } /* :130 */
};

/* ------------------------------------------------------------------------------------------
 * Region view (CompareRegion, src/data_types/compare_region.rs:13-26)
 * ---------------------------------------------------------------------------------------- */
struct Region {
    uint64_t region_id;
    size_t start, end;
    std::vector<Var> truth, query;
};

Region region_view(const avk_region_batch *b, uint64_t r) {
    Region reg;
    reg.region_id = b->region_id ? b->region_id[r] : r;
    reg.start = (size_t)b->start[r];
    reg.end = (size_t)b->end[r];
    auto mk = [&](uint64_t v) {
        Var x;
        x.position = b->var_pos[v];
        x.allele0 = Span(b->allele_bytes + b->a0_off[v], b->a0_len[v]);
        x.allele1 = Span(b->allele_bytes + b->a1_off[v], b->a1_len[v]);
        x.variant_type = b->var_type[v];
        x.zyg = b->var_zyg[v];
        x.raw_allele_space = b->var_raw_space ? b->var_raw_space[v] : std::max(b->a0_len[v], b->a1_len[v]);
        return x;
    };
    for (uint32_t i = 0; i < b->t_cnt[r]; ++i) reg.truth.push_back(mk(b->t_off[r] + i));
    for (uint32_t i = 0; i < b->q_cnt[r]; ++i) reg.query.push_back(mk(b->q_off[r] + i));
    return reg;
}

/* query_optimizer.rs:372-381 — stable sort by position of [truth..., query...] */
struct OrderEntry {
    size_t index;
    bool is_truth;
};
std::vector<OrderEntry> order_variants(const std::vector<Var> &truth, const std::vector<Var> &query) {
    std::vector<OrderEntry> ret;
    for (size_t i = 0; i < truth.size(); ++i) ret.push_back({i, true});
    for (size_t i = 0; i < query.size(); ++i) ret.push_back({i, false});
    std::stable_sort(ret.begin(), ret.end(), [&](const OrderEntry &a, const OrderEntry &b) {
        uint64_t pa = a.is_truth ? truth[a.index].position : query[a.index].position;
        uint64_t pb = b.is_truth ? truth[b.index].position : query[b.index].position;
        return pa < pb;
    });
    return ret;
}

inline bool zyg_is_het(uint8_t z) { return z == AVK_ZYG_UNPHASED_HET || z == AVK_ZYG_PHASED_HET01 || z == AVK_ZYG_PHASED_HET10; }
inline uint8_t zyg_allele_count(uint8_t z) { /* phase_enums.rs:102-112 */
    return zyg_is_het(z) ? 1 : (z == AVK_ZYG_HOM_ALT ? 2 : 0);
}
inline void zyg_decompose(uint8_t z, int *a1, int *a2) { /* phase_enums.rs:91-100 */
    switch (z) {
    case AVK_ZYG_HOM_REF: *a1 = A_REF; *a2 = A_REF; break;
    case AVK_ZYG_UNPHASED_HET:
    case AVK_ZYG_PHASED_HET01: *a1 = A_REF; *a2 = A_ALT; break;
    case AVK_ZYG_PHASED_HET10: *a1 = A_ALT; *a2 = A_REF; break;
    case AVK_ZYG_HOM_ALT: *a1 = A_ALT; *a2 = A_ALT; break;
    default: *a1 = A_UNKNOWN; *a2 = A_UNKNOWN; break;
    }
}

/* ------------------------------------------------------------------------------------------
 * optimize_sequences — src/query_optimizer.rs:166-365
 * ---------------------------------------------------------------------------------------- */
struct ComparisonNode { /* :406-495 */
    uint64_t node_id;
    HaplotypeDWFA hap_dwfa1, hap_dwfa2;
    ComparisonNode(uint64_t id, size_t region_start) : node_id(id), hap_dwfa1(region_start, SIZE_MAX), hap_dwfa2(region_start, SIZE_MAX) {}
    HErr extend_variant(Span reference, bool is_truth, const Var &v, int a1, int a2, bool has_sync, size_t sync) {
        bool s;
        HErr e = hap_dwfa1.extend_variant(reference, is_truth, v, a1, has_sync, sync, &s);
        if (e) return e;
        return hap_dwfa2.extend_variant(reference, is_truth, v, a2, has_sync, sync, &s);
    }
    HErr finalize_dwfas(Span reference, size_t region_end) {
        HErr e = hap_dwfa1.finalize_dwfa(reference, region_end);
        if (e) return e;
        return hap_dwfa2.finalize_dwfa(reference, region_end);
    }
    size_t total_cost() const { return hap_dwfa1.total_cost() + hap_dwfa2.total_cost(); }
    size_t set_alleles() const { return hap_dwfa1.set_alleles(); } /* :478-481 */
};

struct OptimizedHaplotypes { /* :67-92 */
    std::vector<uint8_t> truth_zygosity, query_zygosity;
    Bytes truth_seq1, truth_seq2, query_seq1, query_seq2;
    size_t ed1, ed2, truth_vs1, truth_vs2, query_vs1, query_vs2;
    bool is_exact_match() const { return ed1 + ed2 + truth_vs1 + truth_vs2 + query_vs1 + query_vs2 == 0; }
};

/* :388-402; returns false where the reference panics ("no impl") */
bool convert_alleles_to_zygosity(const std::vector<uint8_t> &a1, const std::vector<uint8_t> &a2, std::vector<uint8_t> *out) {
    if (a1.size() != a2.size()) return false;
    out->clear();
    for (size_t i = 0; i < a1.size(); ++i) {
        if (a1[i] == A_REF && a2[i] == A_ALT) out->push_back(AVK_ZYG_PHASED_HET01);
        else if (a1[i] == A_ALT && a2[i] == A_REF) out->push_back(AVK_ZYG_PHASED_HET10);
        else if (a1[i] == A_ALT && a2[i] == A_ALT) out->push_back(AVK_ZYG_HOM_ALT);
        else return false;
    }
    return true;
}

struct HeapA { /* NodePriority = (Reverse(cost), Reverse(node_id)), :417,:470-475 */
    size_t cost;
    uint64_t id;
    ComparisonNode *node;
};
struct HeapALess { /* std heap is a max-heap: "less" = lower priority = larger (cost, id) */
    bool operator()(const HeapA &a, const HeapA &b) const { return a.cost != b.cost ? a.cost > b.cost : a.id > b.id; }
};

int optimize_sequences(Span reference, size_t start, size_t end, const std::vector<Var> &truth, const std::vector<Var> &query,
                       size_t max_branch_factor, std::vector<OptimizedHaplotypes> *out) {
    if (max_branch_factor == 0) return AVK_ST_BRANCH_FACTOR; /* :177 */
    std::vector<OrderEntry> order = order_variants(truth, query);
    size_t total = order.size();

    uint64_t next_node_id = 0;
    std::vector<HeapA> heap;
    std::vector<std::unique_ptr<ComparisonNode>> owned; /* keeps every allocated node alive until return */
    auto push = [&](ComparisonNode *n) {
        heap.push_back({n->total_cost(), n->node_id, n});
        std::push_heap(heap.begin(), heap.end(), HeapALess());
        t_stats.max_queue_a = std::max<uint64_t>(t_stats.max_queue_a, heap.size());
    };
    owned.emplace_back(new ComparisonNode(next_node_id++, start));
    push(owned.back().get());

    size_t best_ed = SIZE_MAX;
    std::vector<ComparisonNode *> best_results;
    std::vector<size_t> bucket_counts(total + 1, 0);
    uint64_t pops = 0;

    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), HeapALess());
        ComparisonNode *cur = heap.back().node;
        heap.pop_back();
        ++pops;
        if (cur->total_cost() > best_ed) continue; /* :204 */
        size_t order_index = cur->set_alleles();
        if (bucket_counts[order_index] >= max_branch_factor) continue; /* :222 */
        bucket_counts[order_index] += 1;

        if (order_index == total) { /* :227-247 */
            HErr e = cur->finalize_dwfas(reference, end);
            if (e) return e == H_UNKNOWN_ALLELE ? AVK_ST_UNKNOWN_ALLELE : AVK_ST_NO_RESULTS;
            size_t final_cost = cur->total_cost();
            if (final_cost < best_ed) {
                best_ed = final_cost;
                best_results.clear();
                best_results.push_back(cur);
            } else if (final_cost == best_ed) {
                best_results.push_back(cur);
            }
            continue;
        }

        const OrderEntry &oe = order[order_index];
        const Var &v = oe.is_truth ? truth[oe.index] : query[oe.index];
        uint8_t zyg = v.zyg;
        size_t next_var_pos; /* :258-265 */
        if (order_index == total - 1) next_var_pos = end;
        else {
            const OrderEntry &ne = order[order_index + 1];
            next_var_pos = (size_t)(ne.is_truth ? truth[ne.index].position : query[ne.index].position);
        }

        if (zyg_is_het(zyg)) {
            if (!oe.is_truth || zyg == AVK_ZYG_UNPHASED_HET) { /* :269-293 */
                const int ext[2][2] = {{A_REF, A_ALT}, {A_ALT, A_REF}};
                for (int k = 0; k < 2; ++k) {
                    owned.emplace_back(new ComparisonNode(*cur));
                    ComparisonNode *nn = owned.back().get();
                    nn->node_id = next_node_id++;
                    HErr e = nn->extend_variant(reference, oe.is_truth, v, ext[k][0], ext[k][1], true, next_var_pos);
                    if (e) return AVK_ST_UNKNOWN_ALLELE;
                    push(nn);
                }
            } else { /* :294-312 phased truth het: the node is moved, id kept */
                int a1 = zyg == AVK_ZYG_PHASED_HET01 ? A_REF : A_ALT;
                int a2 = zyg == AVK_ZYG_PHASED_HET01 ? A_ALT : A_REF;
                HErr e = cur->extend_variant(reference, oe.is_truth, v, a1, a2, true, next_var_pos);
                if (e) return AVK_ST_UNKNOWN_ALLELE;
                push(cur);
            }
        } else {
            if (zyg != AVK_ZYG_HOM_ALT) return AVK_ST_BAD_ZYGOSITY; /* assert_eq! :315 */
            HErr e = cur->extend_variant(reference, oe.is_truth, v, A_ALT, A_ALT, true, next_var_pos);
            if (e) return AVK_ST_UNKNOWN_ALLELE;
            push(cur);
        }
    }
    t_stats.max_pops_a = std::max(t_stats.max_pops_a, pops);
    t_stats.total_pops_a += pops;

    if (best_results.empty()) return AVK_ST_NO_RESULTS; /* :331 */
    t_stats.max_optima = std::max<uint64_t>(t_stats.max_optima, best_results.size());

    out->clear();
    for (ComparisonNode *bn : best_results) { /* :334-363 */
        OptimizedHaplotypes oh;
        const HaplotypeTracker &t1 = bn->hap_dwfa1.truth_haplotype, &t2 = bn->hap_dwfa2.truth_haplotype;
        const HaplotypeTracker &q1 = bn->hap_dwfa1.query_haplotype, &q2 = bn->hap_dwfa2.query_haplotype;
        if (!convert_alleles_to_zygosity(t1.alleles, t2.alleles, &oh.truth_zygosity)) return AVK_ST_BAD_ZYGOSITY;
        if (!convert_alleles_to_zygosity(q1.alleles, q2.alleles, &oh.query_zygosity)) return AVK_ST_BAD_ZYGOSITY;
        oh.truth_seq1 = t1.sequence;
        oh.truth_seq2 = t2.sequence;
        oh.query_seq1 = q1.sequence;
        oh.query_seq2 = q2.sequence;
        oh.ed1 = bn->hap_dwfa1.edit_distance();
        oh.ed2 = bn->hap_dwfa2.edit_distance();
        oh.truth_vs1 = t1.variant_skip_distance;
        oh.truth_vs2 = t2.variant_skip_distance;
        oh.query_vs1 = q1.variant_skip_distance;
        oh.query_vs2 = q2.variant_skip_distance;
        out->push_back(std::move(oh));
    }
    return AVK_ST_OK;
}

/* ------------------------------------------------------------------------------------------
 * optimize_gt_alleles — src/exact_gt_optimizer.rs:108-357
 * ---------------------------------------------------------------------------------------- */
struct ExactMatchNode { /* :361-479 */
    uint64_t node_id;
    HaplotypeDWFA hap_dwfa;
    size_t num_errors = 0;
    ExactMatchNode(uint64_t id, size_t region_start) : node_id(id), hap_dwfa(region_start, 0) {} /* max ED 0, :380 */
    bool is_exact_match() const { return hap_dwfa.edit_distance() == 0; }
    size_t set_alleles() const { return hap_dwfa.set_alleles(); }
    /* :395-414 — MaxEditDistance is the one allowed error (:482-488) */
    int extend_variant(Span reference, bool is_truth, const Var &v, int allele, bool has_sync, size_t sync, bool is_error, bool *extended) {
        bool s = false;
        HErr e = hap_dwfa.extend_variant(reference, is_truth, v, allele, has_sync, sync, &s);
        if (e == H_D_MAXED) s = false;
        else if (e) return e;
        if (is_error) num_errors += 1;
        *extended = s;
        return 0;
    }
    /* :421-434 */
    int finalize_dwfas(Span reference, size_t region_end) {
        HErr e = hap_dwfa.finalize_dwfa(reference, region_end);
        if (e == H_D_MAXED || e == H_OK) return 0;
        return e;
    }
};

struct HeapB { /* NodePriority = (Reverse(errors), set - errors, Reverse(id)), :372,:452-458 */
    size_t errors, correct;
    uint64_t id;
    ExactMatchNode *node;
};
struct HeapBLess {
    bool operator()(const HeapB &a, const HeapB &b) const {
        if (a.errors != b.errors) return a.errors > b.errors;
        if (a.correct != b.correct) return a.correct < b.correct;
        return a.id > b.id;
    }
};

struct OptimizedAlleles { /* :57-64 */
    std::vector<uint8_t> truth_alleles, query_alleles;
    size_t num_errors;
};

int optimize_gt_alleles(Span reference, size_t start, size_t end, const std::vector<Var> &truth, const std::vector<uint8_t> &truth_alleles,
                        const std::vector<Var> &query, const std::vector<uint8_t> &query_alleles, OptimizedAlleles *out) {
    std::vector<OrderEntry> order = order_variants(truth, query);
    size_t total = order.size();

    uint64_t next_node_id = 0;
    std::vector<HeapB> heap;
    std::vector<std::unique_ptr<ExactMatchNode>> owned;
    auto push = [&](ExactMatchNode *n) {
        heap.push_back({n->num_errors, n->set_alleles() - n->num_errors, n->node_id, n});
        std::push_heap(heap.begin(), heap.end(), HeapBLess());
        t_stats.max_queue_b = std::max<uint64_t>(t_stats.max_queue_b, heap.size());
    };
    owned.emplace_back(new ExactMatchNode(next_node_id++, start));
    push(owned.back().get());

    size_t best_error_count = SIZE_MAX;
    ExactMatchNode *best_result = nullptr;
    size_t min_allele_sync = 0;
    const size_t auto_fail_threshold = 500; /* :160 */
    size_t auto_fail_index = 0, auto_fail_counts = 0;
    uint64_t pops = 0;
    /* the 300 s wall-clock bail (:165,:174-176) is not restated: it is the only nondeterministic
     * element of the path and unreachable under the auto-fail limits */

    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), HeapBLess());
        ExactMatchNode *cur = heap.back().node;
        heap.pop_back();
        ++pops;
        if (cur->num_errors >= best_error_count) continue; /* :169 */

        size_t order_index = cur->set_alleles();
        if (order_index == total) { /* :180-192 */
            int e = cur->finalize_dwfas(reference, end);
            if (e) return AVK_ST_UNKNOWN_ALLELE;
            if (cur->is_exact_match() && cur->num_errors < best_error_count) {
                best_error_count = cur->num_errors;
                best_result = cur;
            }
            continue;
        }
        if (order_index < min_allele_sync) continue; /* :194-197 */
        /* bucket quota is usize::MAX: inert (:152,:200-203) */
        if (cur->hap_dwfa.is_synchronized()) { /* :206-217 */
            min_allele_sync = order_index;
            auto_fail_counts = 0;
            auto_fail_index = min_allele_sync;
        }

        const OrderEntry &oe = order[order_index];
        const Var &v = oe.is_truth ? truth[oe.index] : query[oe.index];
        int current_allele = oe.is_truth ? truth_alleles[oe.index] : query_alleles[oe.index];
        size_t next_var_pos;
        if (order_index == total - 1) next_var_pos = end;
        else {
            const OrderEntry &ne = order[order_index + 1];
            next_var_pos = (size_t)(ne.is_truth ? truth[ne.index].position : query[ne.index].position);
        }

        if (current_allele == A_UNKNOWN) return AVK_ST_UNKNOWN_ALLELE; /* :256 */
        if (current_allele == A_REF) { /* :257-273 node moved, id kept */
            bool success;
            int e = cur->extend_variant(reference, oe.is_truth, v, A_REF, true, next_var_pos, false, &success);
            if (e) return AVK_ST_UNKNOWN_ALLELE;
            if (success && cur->is_exact_match()) push(cur);
        } else { /* :274-306 REF-with-error first, then ALT */
            const int ext_allele[2] = {A_REF, A_ALT};
            const bool ext_error[2] = {true, false};
            for (int k = 0; k < 2; ++k) {
                if (order_index < auto_fail_index && ext_allele[k] != A_REF) continue; /* :282-285 */
                owned.emplace_back(new ExactMatchNode(*cur));
                ExactMatchNode *nn = owned.back().get();
                nn->node_id = next_node_id++;
                bool success;
                int e = nn->extend_variant(reference, oe.is_truth, v, ext_allele[k], true, next_var_pos, ext_error[k], &success);
                if (e) return AVK_ST_UNKNOWN_ALLELE;
                if (success && nn->is_exact_match()) push(nn);
            }
        }

        auto_fail_counts += 1; /* :309-339 */
        if (auto_fail_counts >= auto_fail_threshold) {
            if (auto_fail_index >= total) return AVK_ST_AUTOFAIL_OOB; /* index panic at :312 */
            const OrderEntry &fe = order[auto_fail_index];
            std::vector<HeapB> kept;
            for (const HeapB &h : heap) {
                const std::vector<uint8_t> &al = fe.is_truth ? h.node->hap_dwfa.truth_haplotype.alleles : h.node->hap_dwfa.query_haplotype.alleles;
                int a = fe.index < al.size() ? (int)al[fe.index] : (int)A_REF;
                if (a == A_REF) kept.push_back(h);
            }
            heap.swap(kept);
            std::make_heap(heap.begin(), heap.end(), HeapBLess());
            auto_fail_index += 1;
            auto_fail_counts = 0;
        }
    }
    t_stats.max_pops_b = std::max(t_stats.max_pops_b, pops);
    t_stats.total_pops_b += pops;

    if (!best_result) return AVK_ST_NO_GT_RESULT; /* :345-348 */
    out->truth_alleles = best_result->hap_dwfa.truth_haplotype.alleles;
    out->query_alleles = best_result->hap_dwfa.query_haplotype.alleles;
    out->num_errors = best_result->num_errors;
    return AVK_ST_OK;
}

/* ------------------------------------------------------------------------------------------
 * Metrics containers — src/data_types/{summary_metrics,grouped_metrics,variant_metrics,compare_benchmark}.rs
 * ---------------------------------------------------------------------------------------- */
struct SummaryMetrics {
    uint64_t truth_tp = 0, truth_fn = 0, query_tp = 0, query_fp = 0;
};

struct GroupTypeMetrics { /* grouped_metrics.rs:32-37; m[0] = joint, m[1+t] = variant type t */
    uint64_t m[AVK_N_GROUPS][AVK_N_FIELDS];
    uint16_t present = 0;
    GroupTypeMetrics() { memset(m, 0, sizeof(m)); }

    /* GroupMetrics::add_truth_zygosity, grouped_metrics.rs:183-227 */
    static int group_add_truth(uint64_t *g, uint64_t w, uint8_t exp, uint8_t obs) {
        if (exp == 0) return AVK_ST_VARIANT_METRICS;
        if (exp < obs) return AVK_ST_TRUTH_FP;
        if (exp == obs) {
            g[AVK_F_HAP_TRUTH_TP] += exp;
            g[AVK_F_WHAP_TRUTH_TP] += (uint64_t)exp * w;
            g[AVK_F_GT_TRUTH_TP] += 1;
        } else {
            g[AVK_F_HAP_TRUTH_TP] += obs;
            g[AVK_F_HAP_TRUTH_FN] += (uint64_t)(exp - obs);
            g[AVK_F_WHAP_TRUTH_TP] += (uint64_t)obs * w;
            g[AVK_F_WHAP_TRUTH_FN] += (uint64_t)(exp - obs) * w;
            g[AVK_F_GT_TRUTH_FN] += 1;
            if (obs > 0) g[AVK_F_GT_TRUTH_FN_GT] += 1;
        }
        return 0;
    }
    /* grouped_metrics.rs:45-61 */
    int add_truth_zygosity(const Var &v, uint8_t exp, uint8_t obs) {
        if (exp == 0) return AVK_ST_VARIANT_METRICS;
        uint64_t w = v.alt_ed();
        int e = group_add_truth(m[0], w, exp, obs);
        if (e) return e;
        present |= (uint16_t)(1u << v.variant_type);
        return group_add_truth(m[1 + v.variant_type], w, exp, obs);
    }
    /* GroupMetrics::add_query_zygosity, grouped_metrics.rs:234-249 (exact-shortcut path only) */
    static int group_add_query(uint64_t *g, uint64_t w, uint8_t exp, uint8_t obs) {
        if (exp == 0 || exp != obs) return AVK_ST_VARIANT_METRICS;
        g[AVK_F_HAP_QUERY_TP] += exp;
        g[AVK_F_WHAP_QUERY_TP] += (uint64_t)exp * w;
        g[AVK_F_GT_QUERY_TP] += 1;
        return 0;
    }
    int add_query_zygosity(const Var &v, uint8_t exp, uint8_t obs) {
        uint64_t w = v.alt_ed();
        int e = group_add_query(m[0], w, exp, obs);
        if (e) return e;
        present |= (uint16_t)(1u << v.variant_type);
        return group_add_query(m[1 + v.variant_type], w, exp, obs);
    }
    /* grouped_metrics.rs:87-108 */
    void add_basepair(const SummaryMetrics &s, int vt) {
        uint64_t *g = vt < 0 ? m[0] : m[1 + vt];
        if (vt >= 0) present |= (uint16_t)(1u << vt);
        g[AVK_F_BP_TRUTH_TP] += s.truth_tp;
        g[AVK_F_BP_TRUTH_FN] += s.truth_fn;
        g[AVK_F_BP_QUERY_TP] += s.query_tp;
        g[AVK_F_BP_QUERY_FP] += s.query_fp;
    }
    void add_record_bp(const SummaryMetrics &s, int vt) {
        uint64_t *g = vt < 0 ? m[0] : m[1 + vt];
        if (vt >= 0) present |= (uint16_t)(1u << vt);
        g[AVK_F_RBP_TRUTH_TP] += s.truth_tp;
        g[AVK_F_RBP_TRUTH_FN] += s.truth_fn;
        g[AVK_F_RBP_QUERY_TP] += s.query_tp;
        g[AVK_F_RBP_QUERY_FP] += s.query_fp;
    }
    /* GroupMetrics::add_swap_benchmark, grouped_metrics.rs:268-277 + summary_metrics.rs:33-36,108-111:
     * the query columns are SET from the other benchmark's truth columns */
    static void group_swap(uint64_t *g, const uint64_t *o) {
        g[AVK_F_GT_QUERY_TP] = o[AVK_F_GT_TRUTH_TP];
        g[AVK_F_GT_QUERY_FP] = o[AVK_F_GT_TRUTH_FN];
        g[AVK_F_GT_QUERY_FP_GT] = o[AVK_F_GT_TRUTH_FN_GT];
        g[AVK_F_HAP_QUERY_TP] = o[AVK_F_HAP_TRUTH_TP];
        g[AVK_F_HAP_QUERY_FP] = o[AVK_F_HAP_TRUTH_FN];
        g[AVK_F_WHAP_QUERY_TP] = o[AVK_F_WHAP_TRUTH_TP];
        g[AVK_F_WHAP_QUERY_FP] = o[AVK_F_WHAP_TRUTH_FN];
    }
    /* grouped_metrics.rs:113-120 */
    void add_swap_benchmark(const GroupTypeMetrics &o) {
        group_swap(m[0], o.m[0]);
        for (int t = 0; t < AVK_N_VARIANT_TYPES; ++t)
            if (o.present & (1u << t)) {
                present |= (uint16_t)(1u << t);
                group_swap(m[1 + t], o.m[1 + t]);
            }
    }
};

struct VariantMetrics { /* variant_metrics.rs:26-35 */
    uint8_t is_query, classification, expected, observed;
};
/* variant_metrics.rs:43-71 */
int variant_metrics_new(uint8_t exp, uint8_t obs, VariantMetrics *out) {
    if (exp > 2 || obs > 2) return AVK_ST_VARIANT_METRICS;
    uint8_t c;
    if (exp < obs) c = AVK_CLASS_FP;
    else if (exp == obs) {
        if (exp == 0) return AVK_ST_VARIANT_METRICS;
        c = AVK_CLASS_TP;
    } else c = AVK_CLASS_FN;
    *out = {0, c, exp, obs};
    return 0;
}
/* variant_metrics.rs:77-101 */
VariantMetrics toggle_source(const VariantMetrics &o) {
    VariantMetrics r;
    r.is_query = !o.is_query;
    r.classification = o.classification;
    if (o.classification == AVK_CLASS_FN) r.classification = AVK_CLASS_FP;
    else if (o.classification == AVK_CLASS_FP) r.classification = AVK_CLASS_FN;
    r.expected = o.observed;
    r.observed = o.expected;
    return r;
}

struct CompareBenchmark { /* compare_benchmark.rs:9-33 */
    uint64_t region_id = 0;
    size_t ed_h1 = 0, ed_h2 = 0;
    GroupTypeMetrics group_metrics;
    std::vector<VariantMetrics> truth_variant_data, query_variant_data;
    bool has_sequences = false;
    Bytes seqs[5];
    /* extras for the checker */
    size_t n_optima = 0;
    std::vector<uint8_t> truth_zyg_resolved, query_zyg_resolved;

    /* compare_benchmark.rs:58-67 */
    int add_truth_zygosity(const Var &v, uint8_t exp, uint8_t obs) {
        int e = group_metrics.add_truth_zygosity(v, exp, obs);
        if (e) return e;
        VariantMetrics vm;
        e = variant_metrics_new(exp, obs, &vm);
        if (e) return e;
        truth_variant_data.push_back(vm);
        return 0;
    }
    /* compare_benchmark.rs:74-84 */
    int add_query_zygosity(const Var &v, uint8_t exp, uint8_t obs) {
        int e = group_metrics.add_query_zygosity(v, exp, obs);
        if (e) return e;
        VariantMetrics vm;
        e = variant_metrics_new(exp, obs, &vm);
        if (e) return e;
        query_variant_data.push_back(toggle_source(vm));
        return 0;
    }
    /* compare_benchmark.rs:109-123 */
    void add_swap_benchmark(const CompareBenchmark &other) {
        group_metrics.add_swap_benchmark(other.group_metrics);
        for (const VariantMetrics &vm : other.truth_variant_data) query_variant_data.push_back(toggle_source(vm));
    }
};

/* ------------------------------------------------------------------------------------------
 * waffle_solver.rs helpers
 * ---------------------------------------------------------------------------------------- */
const int SUPPORTED_VARIANT_TYPES[8] = {/* :82-91 */ AVK_VT_SNV, AVK_VT_INSERTION, AVK_VT_DELETION, AVK_VT_INDEL,
                                        AVK_VT_TR_CONTRACTION, AVK_VT_TR_EXPANSION, AVK_VT_SV_DELETION, AVK_VT_SV_INSERTION};

/* :611-658; all values doubled */
SummaryMetrics perform_basepair_compare(Span ref, Span truth, Span query) {
    uint64_t ed_ref_truth = 2 * (uint64_t)wfa_ed(ref, truth);
    uint64_t ed_ref_query = 2 * (uint64_t)wfa_ed(ref, query);
    uint64_t ed_truth_query = 2 * (uint64_t)wfa_ed(truth, query);
    uint64_t tp_shared = (ed_ref_truth + ed_ref_query - ed_truth_query) / 2;
    SummaryMetrics s;
    s.truth_tp = tp_shared;
    s.truth_fn = ed_ref_truth - tp_shared;
    s.query_tp = tp_shared;
    s.query_fp = ed_ref_query - tp_shared;
    return s;
}

/* :726-778 (Reference alleles are skipped entirely: convert_index(0) == 0 always, variants.rs:398-410) */
int generate_allele_sequence(Span reference, size_t start, size_t end, const std::vector<const Var *> &variants,
                             const std::vector<uint8_t> &alleles, Bytes *seq, size_t *failed_ed) {
    size_t cur = start;
    seq->clear();
    *failed_ed = 0;
    for (size_t i = 0; i < variants.size(); ++i) {
        const Var &v = *variants[i];
        int allele = alleles[i];
        if (allele == A_REF) continue;
        size_t vpos = (size_t)v.position;
        if (vpos < cur) { /* :745-753 */
            *failed_ed += v.alt_ed();
            continue;
        }
        seq->insert(seq->end(), reference.p + cur, reference.p + vpos);
        cur = vpos;
        if (allele == A_UNKNOWN) return AVK_ST_UNKNOWN_ALLELE;
        seq->insert(seq->end(), v.allele1.p, v.allele1.p + v.allele1.n);
        cur += v.ref_len();
    }
    if (cur > end) return AVK_ST_INVALID_INPUT; /* get_slice(cur, end) with cur > end panics in the reference */
    seq->insert(seq->end(), reference.p + cur, reference.p + end);
    return 0;
}

/* :685-712 */
int generate_haplotype_sequence(Span reference, size_t start, size_t end, const std::vector<const Var *> &variants,
                                const std::vector<uint8_t> &zygosities, int hap, Bytes *seq, size_t *failed_ed) {
    std::vector<uint8_t> alleles;
    for (uint8_t z : zygosities) {
        if (z == AVK_ZYG_UNKNOWN) return AVK_ST_BAD_ZYGOSITY;
        int a1, a2;
        zyg_decompose(z, &a1, &a2);
        alleles.push_back((uint8_t)(hap == 0 ? a1 : a2));
    }
    return generate_allele_sequence(reference, start, end, variants, alleles, seq, failed_ed);
}

/* :296-327 */
int compare_expected_observed(uint64_t problem_id, size_t ed1, size_t ed2, const std::vector<Var> &variants,
                              const std::vector<uint8_t> &exp_h1, const std::vector<uint8_t> &obs_h1,
                              const std::vector<uint8_t> &exp_h2, const std::vector<uint8_t> &obs_h2, CompareBenchmark *bm) {
    size_t n = variants.size();
    if (exp_h1.size() != n || obs_h1.size() != n || exp_h2.size() != n || obs_h2.size() != n) return AVK_ST_NO_GT_RESULT;
    bm->region_id = problem_id;
    bm->ed_h1 = ed1;
    bm->ed_h2 = ed2;
    for (size_t i = 0; i < n; ++i) {
        uint8_t exp_count = (exp_h1[i] == A_ALT) + (exp_h2[i] == A_ALT);
        uint8_t obs_count = (obs_h1[i] == A_ALT) + (obs_h2[i] == A_ALT);
        if (exp_count < obs_count) return AVK_ST_TRUTH_FP; /* assert! :322 */
        int e = bm->add_truth_zygosity(variants[i], exp_count, obs_count);
        if (e) return e;
    }
    return 0;
}

/* :335-449 */
int add_basepair_stats(const Region &problem, Span reference, CompareBenchmark *bench, const OptimizedHaplotypes &oh) {
    size_t ref_start = problem.start, ref_end = problem.end;
    Span ref_window(reference.p + ref_start, ref_end - ref_start);
    std::vector<const Var *> tv, qv;
    for (const Var &v : problem.truth) tv.push_back(&v);
    for (const Var &v : problem.query) qv.push_back(&v);
    const std::vector<uint8_t> &tz = oh.truth_zygosity, &qz = oh.query_zygosity;

    for (int hap = 0; hap < 2; ++hap) {
        Bytes truth_seq, query_seq;
        size_t truth_ed, query_ed;
        int e = generate_haplotype_sequence(reference, ref_start, ref_end, tv, tz, hap, &truth_seq, &truth_ed);
        if (e) return e;
        e = generate_haplotype_sequence(reference, ref_start, ref_end, qv, qz, hap, &query_seq, &query_ed);
        if (e) return e;
        if (truth_seq != (hap == 0 ? oh.truth_seq1 : oh.truth_seq2)) return AVK_ST_SEQ_MISMATCH; /* :364-367 */
        if (query_seq != (hap == 0 ? oh.query_seq1 : oh.query_seq2)) return AVK_ST_SEQ_MISMATCH;

        if (wfa_ed(truth_seq, query_seq) != (hap == 0 ? oh.ed1 : oh.ed2)) t_stats.incr_mismatch += 1; /* invariant the kernels rely on */
        bench->group_metrics.add_basepair(perform_basepair_compare(ref_window, truth_seq, query_seq), -1);
        SummaryMetrics skip; /* :378-381 */
        skip.truth_fn = 2 * (uint64_t)truth_ed;
        skip.query_fp = 2 * (uint64_t)query_ed;
        bench->group_metrics.add_basepair(skip, -1);

        for (int k = 0; k < 8; ++k) { /* :384-445 */
            int ft = SUPPORTED_VARIANT_TYPES[k];
            std::vector<const Var *> fqv, ftv;
            std::vector<uint8_t> fqz, ftz;
            for (size_t i = 0; i < qv.size(); ++i)
                if (qv[i]->variant_type == ft) {
                    fqv.push_back(qv[i]);
                    fqz.push_back(qz[i]);
                }
            uint64_t query_tp = 0, query_fp = 0;
            if (!fqv.empty()) {
                Bytes filtered_query;
                size_t failed_ed;
                e = generate_haplotype_sequence(reference, ref_start, ref_end, fqv, fqz, hap, &filtered_query, &failed_ed);
                if (e) return e;
                SummaryMetrics fm = perform_basepair_compare(ref_window, truth_seq, filtered_query);
                query_tp = fm.query_tp;
                query_fp = fm.query_fp + 2 * (uint64_t)failed_ed;
            }
            for (size_t i = 0; i < tv.size(); ++i)
                if (tv[i]->variant_type == ft) {
                    ftv.push_back(tv[i]);
                    ftz.push_back(tz[i]);
                }
            uint64_t truth_tp = 0, truth_fn = 0;
            if (!ftv.empty()) {
                Bytes filtered_truth;
                size_t failed_ed;
                e = generate_haplotype_sequence(reference, ref_start, ref_end, ftv, ftz, hap, &filtered_truth, &failed_ed);
                if (e) return e;
                SummaryMetrics fm = perform_basepair_compare(ref_window, filtered_truth, query_seq);
                truth_tp = fm.truth_tp;
                truth_fn = fm.truth_fn + 2 * (uint64_t)failed_ed;
            }
            SummaryMetrics vt;
            vt.truth_tp = truth_tp;
            vt.truth_fn = truth_fn;
            vt.query_tp = query_tp;
            vt.query_fp = query_fp;
            bench->group_metrics.add_basepair(vt, ft); /* called for all 8 types: entries exist even when zero */
        }
    }
    return 0;
}

/* :455-522 — release-build u64 arithmetic wraps (Cargo.toml has no overflow-checks) */
int add_record_basepair_stats(const Region &problem, CompareBenchmark *bench) {
    uint64_t truth_total = 0, query_total = 0;
    uint64_t truth_by_vt[AVK_N_VARIANT_TYPES] = {0}, query_by_vt[AVK_N_VARIANT_TYPES] = {0};
    for (const Var &v : problem.truth) {
        uint64_t c = (uint64_t)zyg_allele_count(v.zyg) * v.raw_allele_space;
        truth_by_vt[v.variant_type] += c;
        truth_total += c;
    }
    for (const Var &v : problem.query) {
        uint64_t c = (uint64_t)zyg_allele_count(v.zyg) * v.raw_allele_space;
        query_by_vt[v.variant_type] += c;
        query_total += c;
    }
    GroupTypeMetrics &gm = bench->group_metrics;
    uint64_t truth_fn = gm.m[0][AVK_F_BP_TRUTH_FN], query_fp = gm.m[0][AVK_F_BP_QUERY_FP];
    uint64_t truth_tp = 2 * truth_total - truth_fn, query_tp = 2 * query_total - query_fp;
    if (truth_tp < gm.m[0][AVK_F_BP_TRUTH_TP]) return AVK_ST_RECORD_BP;
    if (query_tp < gm.m[0][AVK_F_BP_QUERY_TP]) return AVK_ST_RECORD_BP;
    SummaryMetrics s;
    s.truth_tp = truth_tp;
    s.truth_fn = truth_fn;
    s.query_tp = query_tp;
    s.query_fp = query_fp;
    gm.add_record_bp(s, -1);
    uint16_t present = gm.present; /* iterate a snapshot of the map (:501) */
    for (int t = 0; t < AVK_N_VARIANT_TYPES; ++t) {
        if (!(present & (1u << t))) continue;
        uint64_t tfn = gm.m[1 + t][AVK_F_BP_TRUTH_FN], qfp = gm.m[1 + t][AVK_F_BP_QUERY_FP];
        SummaryMetrics v;
        v.truth_tp = 2 * truth_by_vt[t] - tfn;
        v.truth_fn = tfn;
        v.query_tp = 2 * query_by_vt[t] - qfp;
        v.query_fp = qfp;
        gm.add_record_bp(v, t);
    }
    return 0;
}

/* :534-601 hidden --enable-exact-shortcut */
int generate_exact_match(const Region &problem, Span ref_window, const OptimizedHaplotypes &oh, CompareBenchmark *bench) {
    bench->region_id = problem.region_id;
    bench->ed_h1 = 0;
    bench->ed_h2 = 0;
    for (const Var &v : problem.truth) { /* generate_expected_zyg_counts on the RAW zygosity, :783-796 */
        if (v.zyg == AVK_ZYG_UNKNOWN) return AVK_ST_BAD_ZYGOSITY;
        uint8_t ev = zyg_allele_count(v.zyg);
        int e = bench->add_truth_zygosity(v, ev, ev);
        if (e) return e;
    }
    for (const Var &v : problem.query) {
        if (v.zyg == AVK_ZYG_UNKNOWN) return AVK_ST_BAD_ZYGOSITY;
        uint8_t ev = zyg_allele_count(v.zyg);
        int e = bench->add_query_zygosity(v, ev, ev);
        if (e) return e;
    }
    if (oh.truth_seq1 != oh.query_seq1 || oh.truth_seq2 != oh.query_seq2) return AVK_ST_SEQ_MISMATCH; /* :561-562 */
    uint64_t ed1 = wfa_ed(ref_window, oh.truth_seq1), ed2 = wfa_ed(ref_window, oh.truth_seq2);
    SummaryMetrics shared;
    shared.truth_tp = shared.query_tp = 2 * (ed1 + ed2);
    bench->group_metrics.add_basepair(shared, -1);
    for (const Var &v : problem.truth) {
        SummaryMetrics s;
        s.truth_tp = (uint64_t)zyg_allele_count(v.zyg) * 2 * (uint64_t)v.alt_ed();
        bench->group_metrics.add_basepair(s, v.variant_type);
    }
    for (const Var &v : problem.query) {
        SummaryMetrics s;
        s.query_tp = (uint64_t)zyg_allele_count(v.zyg) * 2 * (uint64_t)v.alt_ed();
        bench->group_metrics.add_basepair(s, v.variant_type);
    }
    return 0;
}

/* Conditions under which the reference panics or reads out of bounds before/inside the solver.
 * The host side of the product rejects the same regions with the same status. */
int validate_region(const Region &p, size_t contig_len) {
    if (p.start > p.end || p.end > contig_len) return AVK_ST_INVALID_INPUT;
    if (p.truth.size() + p.query.size() > 60000) return AVK_ST_INVALID_INPUT; /* batch format limit of the product */
    for (int side = 0; side < 2; ++side) {
        const std::vector<Var> &vs = side == 0 ? p.truth : p.query;
        uint64_t last = 0;
        for (const Var &v : vs) {
            if (v.allele0.n == 0 || v.allele1.n == 0) return AVK_ST_INVALID_INPUT;
            if (v.raw_allele_space < std::max(v.allele0.n, v.allele1.n)) return AVK_ST_INVALID_INPUT; /* variants.rs:364-370 */
            if (v.variant_type >= AVK_N_VARIANT_TYPES || v.zyg > AVK_ZYG_HOM_ALT) return AVK_ST_INVALID_INPUT;
            if (v.position < p.start || v.position + v.allele0.n > p.end) return AVK_ST_INVALID_INPUT;
            if (v.position < last) return AVK_ST_INVALID_INPUT;
            last = v.position;
        }
    }
    return 0;
}

/* solve_compare_region — src/waffle_solver.rs:122-284 (stratification lookup stays with the caller) */
int solve_compare_region(const Region &problem, Span reference, const avk_compare_config &cfg, CompareBenchmark *result) {
    int e = validate_region(problem, reference.n);
    if (e) return e;
    size_t ref_start = problem.start, ref_end = problem.end;
    Span ref_seq(reference.p + ref_start, ref_end - ref_start);

    std::vector<OptimizedHaplotypes> all_opt;
    e = optimize_sequences(reference, ref_start, ref_end, problem.truth, problem.query, cfg.max_branch_factor, &all_opt);
    if (e) return e;

    struct Cand {
        const OptimizedHaplotypes *oh;
        OptimizedAlleles h1c, h2c;
        CompareBenchmark truth_stats, query_stats;
    };
    std::vector<std::unique_ptr<Cand>> best_results;
    for (const OptimizedHaplotypes &oh : all_opt) {
        if (cfg.enable_exact_shortcut && oh.is_exact_match()) { /* :171-199 */
            CompareBenchmark exact;
            e = generate_exact_match(problem, ref_seq, oh, &exact);
            if (e) return e;
            exact.n_optima = all_opt.size();
            exact.truth_zyg_resolved = oh.truth_zygosity;
            exact.query_zyg_resolved = oh.query_zygosity;
            if (cfg.enable_sequences) {
                exact.has_sequences = true;
                exact.seqs[0].assign(ref_seq.p, ref_seq.p + ref_seq.n);
                exact.seqs[1] = oh.truth_seq1;
                exact.seqs[2] = oh.truth_seq2;
                exact.seqs[3] = oh.query_seq1;
                exact.seqs[4] = oh.query_seq2;
            }
            *result = std::move(exact);
            return 0;
        }
        std::unique_ptr<Cand> c(new Cand());
        c->oh = &oh;
        std::vector<uint8_t> th1, th2, qh1, qh2; /* :206-211 */
        for (uint8_t z : oh.truth_zygosity) {
            int a1, a2;
            zyg_decompose(z, &a1, &a2);
            th1.push_back((uint8_t)a1);
            th2.push_back((uint8_t)a2);
        }
        for (uint8_t z : oh.query_zygosity) {
            int a1, a2;
            zyg_decompose(z, &a1, &a2);
            qh1.push_back((uint8_t)a1);
            qh2.push_back((uint8_t)a2);
        }
        e = optimize_gt_alleles(reference, ref_start, ref_end, problem.truth, th1, problem.query, qh1, &c->h1c);
        if (e) return e;
        e = optimize_gt_alleles(reference, ref_start, ref_end, problem.truth, th2, problem.query, qh2, &c->h2c);
        if (e) return e;
        e = compare_expected_observed(problem.region_id, oh.ed1, oh.ed2, problem.truth, th1, c->h1c.truth_alleles, th2, c->h2c.truth_alleles, &c->truth_stats);
        if (e) return e;
        if (cfg.enable_sequences) { /* :237-246 */
            c->truth_stats.has_sequences = true;
            c->truth_stats.seqs[0].assign(ref_seq.p, ref_seq.p + ref_seq.n);
            c->truth_stats.seqs[1] = oh.truth_seq1;
            c->truth_stats.seqs[2] = oh.truth_seq2;
            c->truth_stats.seqs[3] = oh.query_seq1;
            c->truth_stats.seqs[4] = oh.query_seq2;
        }
        e = compare_expected_observed(problem.region_id, oh.ed1, oh.ed2, problem.query, qh1, c->h1c.query_alleles, qh2, c->h2c.query_alleles, &c->query_stats);
        if (e) return e;
        best_results.push_back(std::move(c));
    }

    /* :264-265 min_by_key returns the FIRST minimum */
    size_t best = 0;
    for (size_t i = 1; i < best_results.size(); ++i)
        if (best_results[i]->h1c.num_errors + best_results[i]->h2c.num_errors <
            best_results[best]->h1c.num_errors + best_results[best]->h2c.num_errors)
            best = i;
    Cand &w = *best_results[best];
    w.truth_stats.add_swap_benchmark(w.query_stats); /* :269 */
    e = add_basepair_stats(problem, reference, &w.truth_stats, *w.oh); /* :272 */
    if (e) return e;
    e = add_record_basepair_stats(problem, &w.truth_stats); /* :275 */
    if (e) return e;
    w.truth_stats.n_optima = all_opt.size();
    w.truth_stats.truth_zyg_resolved = w.oh->truth_zygosity;
    w.truth_stats.query_zyg_resolved = w.oh->query_zygosity;
    *result = std::move(w.truth_stats);
    return 0;
}

/* merge_solver.rs:211-223 */
int64_t variant_delta_length(const std::vector<Var> &vs) {
    int64_t total = 0;
    for (const Var &v : vs) total += ((int64_t)v.allele1.n - (int64_t)v.allele0.n) * (int64_t)zyg_allele_count(v.zyg);
    return total;
}

void write_result(const avk_region_batch *b, uint64_t r, int status, const CompareBenchmark &bm, avk_result_batch *out) {
    out->status[r] = status;
    bool ok = status == 0;
    if (out->ed_h1) out->ed_h1[r] = ok ? (uint32_t)bm.ed_h1 : 0;
    if (out->ed_h2) out->ed_h2[r] = ok ? (uint32_t)bm.ed_h2 : 0;
    if (out->n_optima) out->n_optima[r] = ok ? (uint32_t)bm.n_optima : 0;
    if (out->type_present) out->type_present[r] = ok ? bm.group_metrics.present : 0;
    if (out->group_metrics) {
        uint32_t *g = out->group_metrics + r * AVK_N_GROUPS * AVK_N_FIELDS;
        for (int i = 0; i < AVK_N_GROUPS; ++i)
            for (int j = 0; j < AVK_N_FIELDS; ++j) g[i * AVK_N_FIELDS + j] = ok ? (uint32_t)bm.group_metrics.m[i][j] : 0;
    }
    for (int side = 0; side < 2; ++side) {
        uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
        uint32_t cnt = side == 0 ? b->t_cnt[r] : b->q_cnt[r];
        const std::vector<VariantMetrics> &vd = side == 0 ? bm.truth_variant_data : bm.query_variant_data;
        const std::vector<uint8_t> &zr = side == 0 ? bm.truth_zyg_resolved : bm.query_zyg_resolved;
        for (uint32_t i = 0; i < cnt; ++i) {
            bool have = ok && i < vd.size();
            if (out->var_expected) out->var_expected[off + i] = have ? vd[i].expected : 0;
            if (out->var_observed) out->var_observed[off + i] = have ? vd[i].observed : 0;
            if (out->var_class) out->var_class[off + i] = have ? vd[i].classification : 0;
            if (out->var_zyg) out->var_zyg[off + i] = (ok && i < zr.size()) ? zr[i] : 0;
        }
    }
    if (out->seq_bytes && out->seq_len) {
        for (int k = 0; k < 5; ++k) {
            uint32_t n = 0;
            if (ok && bm.has_sequences) {
                n = (uint32_t)bm.seqs[k].size();
                if (n > out->seq_stride[r]) n = out->seq_stride[r];
                memcpy(out->seq_bytes + out->seq_off[r] + (uint64_t)k * out->seq_stride[r], bm.seqs[k].data(), n);
            }
            out->seq_len[5 * r + k] = n;
        }
    }
}

thread_local std::vector<OptimizedHaplotypes> t_last_opt;

struct HapNodeSession {
    int n_haps;
    std::vector<HaplotypeDWFA> haps;
    std::vector<Bytes> owned_alleles; /* keeps allele bytes alive */
};

} // namespace

/* ------------------------------------------------------------------------------------------
 * C API
 * ---------------------------------------------------------------------------------------- */
extern "C" {

uint64_t orc_wfa_ed(const uint8_t *a, uint64_t alen, const uint8_t *b, uint64_t blen) { return wfa_ed(Span(a, alen), Span(b, blen)); }
uint64_t orc_edit_distance(const uint8_t *a, uint64_t alen, const uint8_t *b, uint64_t blen) { return edit_distance(Span(a, alen), Span(b, blen)); }

void *orc_dwfa_new(uint64_t max_ed) {
    DWFALite *d = new DWFALite();
    d->max_edit_distance = max_ed == UINT64_MAX ? SIZE_MAX : (size_t)max_ed;
    return d;
}
void orc_dwfa_free(void *h) { delete (DWFALite *)h; }
void *orc_dwfa_clone(const void *h) { return new DWFALite(*(const DWFALite *)h); }
int orc_dwfa_update(void *h, const uint8_t *base, uint64_t blen, const uint8_t *other, uint64_t olen) {
    return ((DWFALite *)h)->update(Span(base, blen), Span(other, olen));
}
int orc_dwfa_finalize(void *h, const uint8_t *base, uint64_t blen, const uint8_t *other, uint64_t olen) {
    return ((DWFALite *)h)->finalize(Span(base, blen), Span(other, olen));
}
uint64_t orc_dwfa_ed(const void *h) { return ((const DWFALite *)h)->edit_distance; }
uint64_t orc_dwfa_wavefront(const void *h, uint64_t *out, uint64_t cap) {
    const DWFALite *d = (const DWFALite *)h;
    for (size_t i = 0; i < d->wavefront.size() && i < cap; ++i) out[i] = d->wavefront[i];
    return d->wavefront.size();
}
int orc_dwfa_equal(const void *a, const void *b) { return *(const DWFALite *)a == *(const DWFALite *)b; }

void *orc_hapnode_new(int n_haps, uint64_t region_start, uint64_t max_ed) {
    HapNodeSession *s = new HapNodeSession();
    s->n_haps = n_haps;
    for (int i = 0; i < n_haps; ++i) s->haps.emplace_back((size_t)region_start, max_ed == UINT64_MAX ? SIZE_MAX : (size_t)max_ed);
    return s;
}
void orc_hapnode_free(void *h) { delete (HapNodeSession *)h; }
int orc_hapnode_extend(void *h, const uint8_t *ref, uint64_t ref_len, int is_truth, uint64_t pos, const uint8_t *a0, uint64_t a0_len,
                       const uint8_t *a1, uint64_t a1_len, int allele_h1, int allele_h2, int64_t sync, int *success_out) {
    HapNodeSession *s = (HapNodeSession *)h;
    s->owned_alleles.emplace_back(a0, a0 + a0_len);
    s->owned_alleles.emplace_back(a1, a1 + a1_len);
    Var v;
    v.position = pos;
    v.allele0 = Span(s->owned_alleles[s->owned_alleles.size() - 2]);
    v.allele1 = Span(s->owned_alleles.back());
    v.variant_type = 0;
    v.zyg = 0;
    v.raw_allele_space = 0;
    bool all = true;
    for (int i = 0; i < s->n_haps; ++i) {
        bool ok = false;
        HErr e = s->haps[i].extend_variant(Span(ref, ref_len), is_truth != 0, v, i == 0 ? allele_h1 : allele_h2, sync >= 0, (size_t)sync, &ok);
        if (e == H_D_MAXED) ok = false; /* ExactMatchNode::extend_variant, exact_gt_optimizer.rs:399-409 */
        else if (e) return e;
        all = all && ok;
    }
    if (success_out) *success_out = all;
    return 0;
}
int orc_hapnode_finalize(void *h, const uint8_t *ref, uint64_t ref_len, uint64_t region_end) {
    HapNodeSession *s = (HapNodeSession *)h;
    for (int i = 0; i < s->n_haps; ++i) {
        HErr e = s->haps[i].finalize_dwfa(Span(ref, ref_len), (size_t)region_end);
        if (e && e != H_D_MAXED) return e;
    }
    return 0;
}
uint64_t orc_hapnode_ed(const void *h, int hap) { return ((const HapNodeSession *)h)->haps[hap].edit_distance(); }
uint64_t orc_hapnode_skip(const void *h, int hap) { return ((const HapNodeSession *)h)->haps[hap].total_skip(); }
uint64_t orc_hapnode_cost(const void *h) {
    const HapNodeSession *s = (const HapNodeSession *)h;
    uint64_t c = 0;
    for (const HaplotypeDWFA &d : s->haps) c += d.total_cost();
    return c;
}
uint64_t orc_hapnode_seq(const void *h, int hap, int is_truth, uint8_t *out, uint64_t cap) {
    const HaplotypeDWFA &d = ((const HapNodeSession *)h)->haps[hap];
    const Bytes &s = is_truth ? d.truth_haplotype.sequence : d.query_haplotype.sequence;
    memcpy(out, s.data(), std::min<uint64_t>(cap, s.size()));
    return s.size();
}
uint64_t orc_hapnode_alleles(const void *h, int hap, int is_truth, uint8_t *out, uint64_t cap) {
    const HaplotypeDWFA &d = ((const HapNodeSession *)h)->haps[hap];
    const std::vector<uint8_t> &a = is_truth ? d.truth_haplotype.alleles : d.query_haplotype.alleles;
    memcpy(out, a.data(), std::min<uint64_t>(cap, a.size()));
    return a.size();
}

int64_t orc_optimize_sequences(const avk_region_batch *batch, uint64_t r, const uint8_t *ref, uint64_t ref_len, uint32_t max_branch_factor,
                               uint32_t cap, uint64_t *ed, uint64_t *skips, uint8_t *truth_zyg, uint8_t *query_zyg) {
    Region reg = region_view(batch, r);
    std::vector<OptimizedHaplotypes> res;
    int e = optimize_sequences(Span(ref, ref_len), reg.start, reg.end, reg.truth, reg.query, max_branch_factor, &res);
    if (e) return -e;
    size_t T = reg.truth.size(), Q = reg.query.size();
    for (size_t k = 0; k < res.size() && k < cap; ++k) {
        ed[2 * k] = res[k].ed1;
        ed[2 * k + 1] = res[k].ed2;
        skips[4 * k] = res[k].truth_vs1;
        skips[4 * k + 1] = res[k].truth_vs2;
        skips[4 * k + 2] = res[k].query_vs1;
        skips[4 * k + 3] = res[k].query_vs2;
        for (size_t i = 0; i < T; ++i) truth_zyg[k * T + i] = res[k].truth_zygosity[i];
        for (size_t i = 0; i < Q; ++i) query_zyg[k * Q + i] = res[k].query_zygosity[i];
    }
    int64_t n = (int64_t)res.size();
    t_last_opt = std::move(res);
    return n;
}
uint64_t orc_last_sequence(uint32_t k, int s, uint8_t *out, uint64_t cap) {
    if (k >= t_last_opt.size()) return 0;
    const OptimizedHaplotypes &oh = t_last_opt[k];
    const Bytes &b = s == 0 ? oh.truth_seq1 : s == 1 ? oh.truth_seq2 : s == 2 ? oh.query_seq1 : oh.query_seq2;
    memcpy(out, b.data(), std::min<uint64_t>(cap, b.size()));
    return b.size();
}

int64_t orc_optimize_gt_alleles(const avk_region_batch *batch, uint64_t r, const uint8_t *ref, uint64_t ref_len, const uint8_t *truth_alleles,
                                const uint8_t *query_alleles, uint8_t *truth_out, uint8_t *query_out) {
    Region reg = region_view(batch, r);
    std::vector<uint8_t> ta(truth_alleles, truth_alleles + reg.truth.size()), qa(query_alleles, query_alleles + reg.query.size());
    OptimizedAlleles oa;
    int e = optimize_gt_alleles(Span(ref, ref_len), reg.start, reg.end, reg.truth, ta, reg.query, qa, &oa);
    if (e) return -e;
    for (size_t i = 0; i < oa.truth_alleles.size(); ++i) truth_out[i] = oa.truth_alleles[i];
    for (size_t i = 0; i < oa.query_alleles.size(); ++i) query_out[i] = oa.query_alleles[i];
    return (int64_t)oa.num_errors;
}

void orc_basepair_compare(const uint8_t *ref, uint64_t rl, const uint8_t *t, uint64_t tl, const uint8_t *q, uint64_t ql, uint64_t out[4]) {
    SummaryMetrics s = perform_basepair_compare(Span(ref, rl), Span(t, tl), Span(q, ql));
    out[0] = s.truth_tp;
    out[1] = s.truth_fn;
    out[2] = s.query_tp;
    out[3] = s.query_fp;
}

/* small windows onto the metric containers for the data-type golden tests */
int orc_group_add_truth(uint64_t g[AVK_N_FIELDS], uint64_t weight, uint8_t expected, uint8_t observed) {
    return GroupTypeMetrics::group_add_truth(g, weight, expected, observed);
}
int orc_group_add_query(uint64_t g[AVK_N_FIELDS], uint64_t weight, uint8_t expected, uint8_t observed) {
    return GroupTypeMetrics::group_add_query(g, weight, expected, observed);
}
void orc_group_swap(uint64_t g[AVK_N_FIELDS], const uint64_t other[AVK_N_FIELDS]) { GroupTypeMetrics::group_swap(g, other); }
int orc_variant_metrics(uint8_t expected, uint8_t observed, uint8_t out[3], uint8_t toggled[3]) {
    VariantMetrics vm;
    int e = variant_metrics_new(expected, observed, &vm);
    if (e) return e;
    out[0] = vm.expected;
    out[1] = vm.observed;
    out[2] = vm.classification;
    VariantMetrics t = toggle_source(vm);
    toggled[0] = t.expected;
    toggled[1] = t.observed;
    toggled[2] = t.classification;
    return 0;
}

/* generate_haplotype_sequence (waffle_solver.rs:685-712) over the truth (side 0) or query (side 1)
 * variants of region r with the given zygosities; hap 0/1. returns the length or -status */
int64_t orc_generate_haplotype_sequence(const avk_region_batch *batch, uint64_t r, int side, const uint8_t *ref, uint64_t ref_len,
                                        const uint8_t *zygosities, int hap, uint8_t *out, uint64_t cap, uint64_t *failed_ed) {
    Region reg = region_view(batch, r);
    const std::vector<Var> &vs = side == 0 ? reg.truth : reg.query;
    std::vector<const Var *> pv;
    for (const Var &v : vs) pv.push_back(&v);
    std::vector<uint8_t> z(zygosities, zygosities + vs.size());
    Bytes seq;
    size_t fe = 0;
    int e = generate_haplotype_sequence(Span(ref, ref_len), reg.start, reg.end, pv, z, hap, &seq, &fe);
    if (e) return -e;
    memcpy(out, seq.data(), std::min<uint64_t>(cap, seq.size()));
    *failed_ed = fe;
    return (int64_t)seq.size();
}

int orc_compare_batch(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
                      const avk_compare_config *cfg, avk_result_batch *out, int threads) {
    if (!batch || !cfg || !out || !out->status) return AVK_E_ARG;
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> next(0);
    std::vector<uint64_t> tallies((size_t)threads * AVK_TALLY_LEN, 0);
    {
        std::lock_guard<std::mutex> lk(g_stats_mutex);
        g_stats = Stats();
    }
    auto worker = [&](int tid) {
        t_stats = Stats();
        uint64_t *tally = tallies.data() + (size_t)tid * AVK_TALLY_LEN;
        const uint64_t chunk = 64;
        for (;;) {
            uint64_t r0 = next.fetch_add(chunk);
            if (r0 >= batch->n_regions) break;
            uint64_t r1 = std::min(batch->n_regions, r0 + chunk);
            for (uint64_t r = r0; r < r1; ++r) {
                CompareBenchmark bm;
                int st;
                uint32_t c = batch->contig_idx ? batch->contig_idx[r] : 0;
                if (c >= n_contigs) st = AVK_ST_INVALID_INPUT;
                else {
                    Region reg = region_view(batch, r);
                    st = solve_compare_region(reg, Span(refs[c], ref_lens[c]), *cfg, &bm);
                }
                write_result(batch, r, st, bm, out);
                if (st == 0) { /* SummaryWriter::add_comparison_benchmark, writers/summary.rs:146-158 */
                    for (int i = 0; i < AVK_N_GROUPS; ++i)
                        for (int j = 0; j < AVK_N_FIELDS; ++j) tally[i * AVK_N_FIELDS + j] += bm.group_metrics.m[i][j];
                    tally[AVK_TALLY_SOLVED] += 1;
                } else tally[AVK_TALLY_ERRORS] += 1;
            }
        }
        std::lock_guard<std::mutex> lk(g_stats_mutex);
        g_stats.merge(t_stats);
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back(worker, t);
        for (std::thread &t : ts) t.join();
    }
    if (out->tally) {
        for (int i = 0; i < AVK_TALLY_LEN; ++i) {
            uint64_t s = 0;
            for (int t = 0; t < threads; ++t) s += tallies[(size_t)t * AVK_TALLY_LEN + i];
            out->tally[i] = s;
        }
    }
    return 0;
}

/* Timed CPU baseline for bench.py: `reps` passes of solve_compare_region over the whole batch on
 * `threads` workers (dynamic 64-region chunks, like the rayon loop of main.rs:251-268).  The clock
 * starts when every worker is ready and stops when the last one finishes; results are discarded
 * except for a checksum of the statuses + joint tallies so the work cannot be optimised away. */
int orc_bench(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
              const avk_compare_config *cfg, int threads, int reps, double *seconds, uint64_t *checksum) {
    if (!batch || !cfg || !seconds) return AVK_E_ARG;
    if (threads < 1) threads = 1;
    if (reps < 1) reps = 1;
    const uint64_t total = batch->n_regions * (uint64_t)reps;
    std::atomic<uint64_t> next(0), sum(0);
    std::atomic<int> ready(0);
    std::atomic<bool> go(false);
    auto worker = [&]() {
        uint64_t local = 0;
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        const uint64_t chunk = 64;
        for (;;) {
            uint64_t i0 = next.fetch_add(chunk);
            if (i0 >= total) break;
            uint64_t i1 = std::min(total, i0 + chunk);
            for (uint64_t i = i0; i < i1; ++i) {
                const uint64_t r = i % batch->n_regions;
                CompareBenchmark bm;
                uint32_t c = batch->contig_idx ? batch->contig_idx[r] : 0;
                int st = AVK_ST_INVALID_INPUT;
                if (c < n_contigs) {
                    Region reg = region_view(batch, r);
                    st = solve_compare_region(reg, Span(refs[c], ref_lens[c]), *cfg, &bm);
                }
                local += (uint64_t)st * 1000003u + bm.group_metrics.m[0][AVK_F_GT_TRUTH_TP] + 3 * bm.group_metrics.m[0][AVK_F_BP_TRUTH_TP];
            }
        }
        sum.fetch_add(local);
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t) ts.emplace_back(worker);
    while (ready.load() < threads) std::this_thread::yield();
    auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (std::thread &t : ts) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (checksum) *checksum = sum.load();
    return 0;
}

/* per-region search statistics (sizing / tuning aid): out = pops A, max queue A, pops B (all runs), max queue B, optima, max ed */
int orc_region_stats(const avk_region_batch *batch, uint64_t r, const uint8_t *ref, uint64_t ref_len, const avk_compare_config *cfg, uint64_t out[6]) {
    t_stats = Stats();
    Region reg = region_view(batch, r);
    CompareBenchmark bm;
    int st = solve_compare_region(reg, Span(ref, ref_len), *cfg, &bm);
    out[0] = t_stats.total_pops_a;
    out[1] = t_stats.max_queue_a;
    out[2] = t_stats.total_pops_b;
    out[3] = t_stats.max_queue_b;
    out[4] = t_stats.max_optima;
    out[5] = t_stats.max_ed;
    return st;
}

int orc_optimize_pairs_batch(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
                             uint32_t max_branch_factor, int32_t *status, uint8_t *is_exact_match, int threads) {
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> next(0);
    auto worker = [&]() {
        for (;;) {
            uint64_t r = next.fetch_add(1);
            if (r >= batch->n_regions) break;
            uint32_t c = batch->contig_idx ? batch->contig_idx[r] : 0;
            is_exact_match[r] = 0;
            if (c >= n_contigs) {
                status[r] = AVK_ST_INVALID_INPUT;
                continue;
            }
            Region reg = region_view(batch, r);
            Span reference(refs[c], ref_lens[c]);
            int e = validate_region(reg, reference.n);
            if (e) {
                status[r] = e;
                continue;
            }
            bool unknown = false; /* variant_delta_length bails on an Unknown zygosity, merge_solver.rs:216 */
            for (const Var &v : reg.truth) unknown = unknown || v.zyg == AVK_ZYG_UNKNOWN;
            for (const Var &v : reg.query) unknown = unknown || v.zyg == AVK_ZYG_UNKNOWN;
            if (unknown) {
                status[r] = AVK_ST_BAD_ZYGOSITY;
                continue;
            }
            if (variant_delta_length(reg.truth) != variant_delta_length(reg.query)) { /* merge_solver.rs:135,:147 */
                status[r] = 0;
                continue;
            }
            std::vector<OptimizedHaplotypes> res;
            e = optimize_sequences(reference, reg.start, reg.end, reg.truth, reg.query, max_branch_factor, &res);
            status[r] = e;
            if (!e) is_exact_match[r] = res[0].is_exact_match() ? 1 : 0;
        }
    };
    if (threads == 1) worker();
    else {
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back(worker);
        for (std::thread &t : ts) t.join();
    }
    return 0;
}

void orc_last_stats(uint64_t out[16]) {
    std::lock_guard<std::mutex> lk(g_stats_mutex);
    memset(out, 0, 16 * sizeof(uint64_t));
    out[0] = g_stats.max_pops_a;
    out[1] = g_stats.max_queue_a;
    out[2] = g_stats.max_pops_b;
    out[3] = g_stats.max_queue_b;
    out[4] = g_stats.max_ed;
    out[5] = g_stats.max_optima;
    out[6] = g_stats.total_pops_a;
    out[7] = g_stats.total_pops_b;
    out[8] = g_stats.total_wfa;
    out[9] = g_stats.incr_mismatch;
    out[10] = g_stats.byte_compares;
}

} /* extern "C" */
