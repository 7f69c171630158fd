/*
 * avk_shard_host.inl — the multi-GPU rule of the product, host side: which shard a region belongs to (avk_region_shard of the public header: ONE rule for the
 * C++ tools and for aardvark_amd/dist.py), a packed batch cut into the shard a rank owns, its results scattered back, and the job tally summed over the ranks by
 * one RCCL all-reduce (SummaryWriter::add_comparison_benchmark over all regions, src/writers/summary.rs:146-163: integer sums, order-independent).
 * Regions are independent (src/data_types/compare_region.rs:9-11, the rayon loop of src/main.rs:251-268): there is no data-path collective.
 */
#ifndef AVK_SHARD_HOST_INL
#define AVK_SHARD_HOST_INL

#include <dlfcn.h>

struct avk_packed_shard {
    avk_packed_batch b;
    std::vector<uint64_t> index;      /* the shard's k-th region is region index[k] of the whole batch */
    std::vector<uint64_t> whole_call; /* ... and its calls start at call whole_call[k] of the whole batch */
    std::vector<uint16_t> contig_idx, len, rel;
    std::vector<uint32_t> start, raw;
    std::vector<uint8_t> t_cnt, q_cnt, tz, a0, a1, alleles;
};

struct avk_packed_multi_shard {
    avk_packed_multi_batch b;
    std::vector<uint64_t> index; /* the shard's j-th region is region index[j] of the whole batch */
    std::vector<uint16_t> contig_idx, len, rel;
    std::vector<uint32_t> start, raw;
    std::vector<uint8_t> in_cnt, tz, a0, a1, alleles;
};

extern "C" {

int avk_packed_shard_make(const avk_packed_batch *whole, const uint64_t *region_id, uint64_t first_id, uint32_t rank, uint32_t world, avk_packed_shard **out) {
    if (!whole || !out || world == 0 || rank >= world) return AVK_E_ARG;
    *out = nullptr;
    const uint64_t n = whole->n_regions, nv = whole->n_variants;
    if (n && (!whole->start || !whole->len || !whole->t_cnt || !whole->q_cnt)) return AVK_E_ARG;
    if (nv && (!whole->var_rel_pos || !whole->var_type_zyg || !whole->a0_len || !whole->a1_len || !whole->allele_bytes)) return AVK_E_ARG;
    avk_packed_shard *s = new avk_packed_shard();
    memset(&s->b, 0, sizeof(s->b));
    /* the packed form's offsets are running sums: where every region's calls and every call's alleles start */
    std::vector<uint64_t> v_first(n + 1, 0), a_first(nv + 1, 0);
    for (uint64_t r = 0; r < n; ++r) v_first[r + 1] = v_first[r] + whole->t_cnt[r] + whole->q_cnt[r];
    if (v_first[n] != nv) {
        delete s;
        return AVK_E_ARG;
    }
    for (uint64_t v = 0; v < nv; ++v) a_first[v + 1] = a_first[v] + whole->a0_len[v] + whole->a1_len[v];
    if (a_first[nv] != whole->allele_bytes_len) {
        delete s;
        return AVK_E_ARG;
    }
    for (uint64_t r = 0; r < n; ++r)
        if (avk_region_shard(region_id ? region_id[r] : first_id + r, world) == rank) s->index.push_back(r);
    const uint64_t m = s->index.size();
    uint64_t mv = 0, ma = 0;
    for (uint64_t k = 0; k < m; ++k) {
        const uint64_t r = s->index[k];
        mv += v_first[r + 1] - v_first[r];
        ma += a_first[v_first[r + 1]] - a_first[v_first[r]];
    }
    s->whole_call.resize(m + 1);
    s->start.resize(m + 1), s->len.resize(m + 1), s->t_cnt.resize(m + 1), s->q_cnt.resize(m + 1);
    if (whole->contig_idx) s->contig_idx.resize(m + 1);
    s->rel.resize(mv + 1), s->tz.resize(mv + 1), s->a0.resize(mv + 1), s->a1.resize(mv + 1), s->alleles.resize(ma + 1);
    if (whole->var_raw_space) s->raw.resize(mv + 1);
    uint64_t at_v = 0, at_a = 0;
    for (uint64_t k = 0; k < m; ++k) {
        const uint64_t r = s->index[k], v0 = v_first[r], cnt = v_first[r + 1] - v0, ab = a_first[v0 + cnt] - a_first[v0];
        s->whole_call[k] = v0;
        s->start[k] = whole->start[r], s->len[k] = whole->len[r], s->t_cnt[k] = whole->t_cnt[r], s->q_cnt[k] = whole->q_cnt[r];
        if (whole->contig_idx) s->contig_idx[k] = whole->contig_idx[r];
        memcpy(s->rel.data() + at_v, whole->var_rel_pos + v0, cnt * 2);
        memcpy(s->tz.data() + at_v, whole->var_type_zyg + v0, cnt);
        memcpy(s->a0.data() + at_v, whole->a0_len + v0, cnt);
        memcpy(s->a1.data() + at_v, whole->a1_len + v0, cnt);
        if (whole->var_raw_space) memcpy(s->raw.data() + at_v, whole->var_raw_space + v0, cnt * 4);
        memcpy(s->alleles.data() + at_a, whole->allele_bytes + a_first[v0], ab);
        at_v += cnt, at_a += ab;
    }
    s->whole_call[m] = nv;
    s->b.n_regions = m, s->b.n_variants = mv, s->b.allele_bytes_len = ma;
    s->b.contig_idx = whole->contig_idx ? s->contig_idx.data() : nullptr;
    s->b.start = s->start.data(), s->b.len = s->len.data(), s->b.t_cnt = s->t_cnt.data(), s->b.q_cnt = s->q_cnt.data();
    s->b.var_rel_pos = s->rel.data(), s->b.var_type_zyg = s->tz.data(), s->b.a0_len = s->a0.data(), s->b.a1_len = s->a1.data();
    s->b.var_raw_space = whole->var_raw_space ? s->raw.data() : nullptr;
    s->b.allele_bytes = s->alleles.data();
    *out = s;
    return AVK_E_OK;
}

const avk_packed_batch *avk_packed_shard_batch(const avk_packed_shard *s) { return s ? &s->b : nullptr; }

uint64_t avk_packed_shard_regions(const avk_packed_shard *s, const uint64_t **index_in_whole) {
    if (!s) return 0;
    if (index_in_whole) *index_in_whole = s->index.data();
    return s->index.size();
}

/* the shard's results into the arrays of the whole batch: per-region arrays at the regions' indices there, per-call arrays at the regions' calls there; the
 * tally is the caller's to add (or avk_tally_allreduce's) */
int avk_packed_shard_scatter(const avk_packed_shard *s, const avk_result_batch *from, avk_result_batch *to) {
    if (!s || !from || !to) return AVK_E_ARG;
    const uint64_t m = s->index.size();
    uint64_t at_v = 0;
    for (uint64_t k = 0; k < m; ++k) {
        const uint64_t r = s->index[k], cnt = (uint64_t)s->t_cnt[k] + s->q_cnt[k], v0 = s->whole_call[k];
        if (from->status && to->status) to->status[r] = from->status[k];
        if (from->ed_h1 && to->ed_h1) to->ed_h1[r] = from->ed_h1[k];
        if (from->ed_h2 && to->ed_h2) to->ed_h2[r] = from->ed_h2[k];
        if (from->n_optima && to->n_optima) to->n_optima[r] = from->n_optima[k];
        if (from->type_present && to->type_present) to->type_present[r] = from->type_present[k];
        if (from->region_packed && to->region_packed) to->region_packed[r] = from->region_packed[k];
        if (from->var_expected && to->var_expected) memcpy(to->var_expected + v0, from->var_expected + at_v, cnt);
        if (from->var_observed && to->var_observed) memcpy(to->var_observed + v0, from->var_observed + at_v, cnt);
        if (from->var_class && to->var_class) memcpy(to->var_class + v0, from->var_class + at_v, cnt);
        if (from->var_zyg && to->var_zyg) memcpy(to->var_zyg + v0, from->var_zyg + at_v, cnt);
        if (from->var_packed && to->var_packed) memcpy(to->var_packed + v0, from->var_packed + at_v, cnt);
        at_v += cnt;
    }
    return AVK_E_OK;
}

void avk_packed_shard_free(avk_packed_shard *s) { delete s; }

/* ---- merge (solve_merge_region): regions are mapped like compare regions (src/main.rs:463-478), so a packed multi-region batch is cut by the same rule ---- */

int avk_packed_multi_shard_make(const avk_packed_multi_batch *whole, const uint64_t *region_id, uint64_t first_id, uint32_t rank, uint32_t world,
                                avk_packed_multi_shard **out) {
    if (!whole || !out || world == 0 || rank >= world || whole->n_inputs < 2 || whole->n_inputs > 64) return AVK_E_ARG;
    *out = nullptr;
    const uint64_t n = whole->n_regions, nv = whole->n_variants, k = whole->n_inputs;
    if (n && (!whole->start || !whole->len || !whole->in_cnt)) return AVK_E_ARG;
    if (nv && (!whole->var_rel_pos || !whole->var_type_zyg || !whole->a0_len || !whole->a1_len || !whole->allele_bytes)) return AVK_E_ARG;
    std::vector<uint64_t> v_first(n + 1, 0), a_first(nv + 1, 0);
    for (uint64_t r = 0; r < n; ++r) {
        uint64_t c = 0;
        for (uint64_t i = 0; i < k; ++i) c += whole->in_cnt[r * k + i];
        v_first[r + 1] = v_first[r] + c;
    }
    if (v_first[n] != nv) return AVK_E_ARG;
    for (uint64_t v = 0; v < nv; ++v) a_first[v + 1] = a_first[v] + whole->a0_len[v] + whole->a1_len[v];
    if (a_first[nv] != whole->allele_bytes_len) return AVK_E_ARG;
    avk_packed_multi_shard *s = new avk_packed_multi_shard();
    memset(&s->b, 0, sizeof(s->b));
    for (uint64_t r = 0; r < n; ++r)
        if (avk_region_shard(region_id ? region_id[r] : first_id + r, world) == rank) s->index.push_back(r);
    const uint64_t m = s->index.size();
    uint64_t mv = 0, ma = 0;
    for (uint64_t j = 0; j < m; ++j) {
        const uint64_t r = s->index[j];
        mv += v_first[r + 1] - v_first[r];
        ma += a_first[v_first[r + 1]] - a_first[v_first[r]];
    }
    s->start.resize(m + 1), s->len.resize(m + 1), s->in_cnt.resize(m * k + 1);
    if (whole->contig_idx) s->contig_idx.resize(m + 1);
    s->rel.resize(mv + 1), s->tz.resize(mv + 1), s->a0.resize(mv + 1), s->a1.resize(mv + 1), s->alleles.resize(ma + 1);
    if (whole->var_raw_space) s->raw.resize(mv + 1);
    uint64_t at_v = 0, at_a = 0;
    for (uint64_t j = 0; j < m; ++j) {
        const uint64_t r = s->index[j], v0 = v_first[r], cnt = v_first[r + 1] - v0, ab = a_first[v0 + cnt] - a_first[v0];
        s->start[j] = whole->start[r], s->len[j] = whole->len[r];
        memcpy(s->in_cnt.data() + j * k, whole->in_cnt + r * k, k);
        if (whole->contig_idx) s->contig_idx[j] = whole->contig_idx[r];
        memcpy(s->rel.data() + at_v, whole->var_rel_pos + v0, cnt * 2);
        memcpy(s->tz.data() + at_v, whole->var_type_zyg + v0, cnt);
        memcpy(s->a0.data() + at_v, whole->a0_len + v0, cnt);
        memcpy(s->a1.data() + at_v, whole->a1_len + v0, cnt);
        if (whole->var_raw_space) memcpy(s->raw.data() + at_v, whole->var_raw_space + v0, cnt * 4);
        memcpy(s->alleles.data() + at_a, whole->allele_bytes + a_first[v0], ab);
        at_v += cnt, at_a += ab;
    }
    s->b.n_regions = m, s->b.n_inputs = (uint32_t)k, s->b.n_variants = mv, s->b.allele_bytes_len = ma;
    s->b.contig_idx = whole->contig_idx ? s->contig_idx.data() : nullptr;
    s->b.start = s->start.data(), s->b.len = s->len.data(), s->b.in_cnt = s->in_cnt.data();
    s->b.var_rel_pos = s->rel.data(), s->b.var_type_zyg = s->tz.data(), s->b.a0_len = s->a0.data(), s->b.a1_len = s->a1.data();
    s->b.var_raw_space = whole->var_raw_space ? s->raw.data() : nullptr;
    s->b.allele_bytes = s->alleles.data();
    *out = s;
    return AVK_E_OK;
}

const avk_packed_multi_batch *avk_packed_multi_shard_batch(const avk_packed_multi_shard *s) { return s ? &s->b : nullptr; }

uint64_t avk_packed_multi_shard_regions(const avk_packed_multi_shard *s, const uint64_t **index_in_whole) {
    if (!s) return 0;
    if (index_in_whole) *index_in_whole = s->index.data();
    return s->index.size();
}

/* the shard's three result arrays into the whole batch's, region by region (a merge has no per-call outputs) */
int avk_packed_multi_shard_scatter(const avk_packed_multi_shard *s, const int32_t *status, const uint8_t *classification, const uint64_t *members,
                                   int32_t *whole_status, uint8_t *whole_classification, uint64_t *whole_members) {
    if (!s) return AVK_E_ARG;
    const uint64_t m = s->index.size();
    for (uint64_t j = 0; j < m; ++j) {
        const uint64_t r = s->index[j];
        if (status && whole_status) whole_status[r] = status[j];
        if (classification && whole_classification) whole_classification[r] = classification[j];
        if (members && whole_members) whole_members[r] = members[j];
    }
    return AVK_E_OK;
}

void avk_packed_multi_shard_free(avk_packed_multi_shard *s) { delete s; }

/* MergeSummaryWriter's map (src/writers/merge_summary.rs:12-18,57-81) as one dense block of sums, so that the ranks of a sharded merge add theirs up with one
 * all-reduce.  A key is (merge reason with its indices, variant type, input): the reasons are numbered in the order of the reference's derive(Ord) —
 * Different, NoConflict{mask 0 .. 2^k - 1}, MajorityAgree{mask}, ConflictSelection{index 0 .. k - 1}, BasepairIdentical — which is NOT the order of its index
 * lists (they compare lexicographically; the writer sorts, avf_write_merge_summary_counts).  Entry = ((reason * 12 + type) * k + input) * 2 + (0 pass | 1 fail). */
uint64_t avk_merge_counts_len(uint32_t n_inputs) {
    if (n_inputs < 2 || n_inputs > AVK_MERGE_COUNTS_MAX_INPUTS) return 0;
    const uint64_t reasons = 2 + 2 * (1ull << n_inputs) + n_inputs;
    return reasons * AVK_N_VARIANT_TYPES * n_inputs * 2;
}

uint32_t avk_merge_counts_reason(uint32_t n_inputs, uint8_t classification, uint64_t members) {
    const uint32_t masks = 1u << n_inputs;
    switch (classification) {
    case AVK_MERGE_DIFFERENT: return 0;
    case AVK_MERGE_NO_CONFLICT: return 1 + (uint32_t)(members & (masks - 1));
    case AVK_MERGE_MAJORITY_AGREE: return 1 + masks + (uint32_t)(members & (masks - 1));
    case AVK_MERGE_CONFLICT_SELECTION: return 1 + 2 * masks + (uint32_t)(members < n_inputs ? members : 0);
    default: return 1 + 2 * masks + n_inputs; /* AVK_MERGE_IDENTICAL */
    }
}

int avk_merge_counts(const avk_packed_multi_batch *b, const int32_t *status, const uint8_t *classification, const uint64_t *members, uint64_t *counts) {
    if (!b || !status || !classification || !members || !counts) return AVK_E_ARG;
    const uint32_t k = b->n_inputs;
    if (avk_merge_counts_len(k) == 0) return AVK_E_ARG;
    uint64_t v = 0;
    for (uint64_t r = 0; r < b->n_regions; ++r) {
        const uint8_t cls = classification[r];
        if (status[r] == 0 && (cls > AVK_MERGE_CONFLICT_SELECTION || (cls == AVK_MERGE_CONFLICT_SELECTION && members[r] >= k))) return AVK_E_ARG;
        const uint64_t reason = avk_merge_counts_reason(k, cls, members[r]);
        for (uint32_t i = 0; i < k; ++i) {
            const uint32_t cnt = b->in_cnt[r * k + i];
            if (status[r] == 0) { /* unsolved regions are not added (the reference logs the error and moves on, src/main.rs:481-497) */
                /* is_passing (:61-72): every input of a BasepairIdentical region, the listed ones otherwise */
                const bool passing = cls == AVK_MERGE_IDENTICAL || (cls == AVK_MERGE_CONFLICT_SELECTION ? members[r] == i : cls != AVK_MERGE_DIFFERENT && (members[r] >> i & 1));
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t vt = b->var_type_zyg[v + j] & 15;
                    if (vt >= AVK_N_VARIANT_TYPES) return AVK_E_ARG;
                    counts[((reason * AVK_N_VARIANT_TYPES + vt) * k + i) * 2 + (passing ? 0 : 1)] += 1;
                }
            }
            v += cnt;
        }
    }
    return v == b->n_variants ? AVK_E_OK : AVK_E_ARG;
}

namespace {
typedef int (*avk_nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
/* RCCL is not linked: the caller that made the communicator has the library in the process and its ncclAllReduce is looked up there, once (the tool's rank
 * threads all come through here).  ncclUint64 = 5 and ncclSum = 0 are RCCL's nccl.h values (ncclDataType_t / ncclRedOp_t, unchanged since NCCL 2.0); the
 * loaded library has to be a 2.x for them to hold, which ncclGetVersion confirms. */
avk_nccl_allreduce_fn nccl_allreduce_lookup(std::string &why) {
    static std::string err;
    static const avk_nccl_allreduce_fn fn = [] {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char *e = dlerror();
            err = std::string("RCCL is not in the process and cannot be loaded: ") + (e ? e : "?");
            return (avk_nccl_allreduce_fn) nullptr;
        }
        typedef int (*version_fn)(int *);
        version_fn ver = (version_fn)dlsym(h, "ncclGetVersion");
        int v = 0;
        if (!ver || ver(&v) != 0 || v < 2000 || v >= 30000) { /* 2.x.y is reported as 2xxyy (or 2xyy before 2.9) */
            err = "librccl.so does not report an NCCL 2.x interface (ncclGetVersion " + std::to_string(v) + ")";
            return (avk_nccl_allreduce_fn) nullptr;
        }
        avk_nccl_allreduce_fn f = (avk_nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!f) err = "ncclAllReduce not found in librccl.so";
        return f;
    }();
    why = err;
    return fn;
}
} // namespace

/* `n` 64-bit sums of this rank added up over the ranks of `nccl_comm` (an ncclComm_t of RCCL; every rank calls with its own context and communicator): one
 * ncclAllReduce on the context's stream.  The collective of a multi-GPU compare (the 13 x 22 + 2 tally, avk_tally_allreduce) and of a multi-GPU merge (the
 * summary counters, avk_merge_counts). */
int avk_counts_allreduce(avk_ctx *ctx, void *nccl_comm, uint64_t *counts, uint64_t n) {
    if (!ctx || !nccl_comm || !counts || n == 0) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    std::string why;
    const avk_nccl_allreduce_fn fn = nccl_allreduce_lookup(why);
    if (!fn) return fail(ctx, AVK_E_STATE, "%s", why.c_str());
    uint64_t *d = nullptr;
    const int rc = pool_alloc(ctx, (void **)&d, (size_t)n * sizeof(uint64_t));
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(d, counts, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    int nrc = 0;
    if (e == hipSuccess) nrc = fn(d, d, (size_t)n, /* ncclUint64 */ 5, /* ncclSum */ 0, nccl_comm, ctx->stream);
    if (e == hipSuccess && nrc == 0) e = hipMemcpyAsync(counts, d, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    pool_release(ctx, d);
    if (nrc != 0) return fail(ctx, AVK_E_HIP, "ncclAllReduce failed (%d)", nrc);
    if (e != hipSuccess) return fail(ctx, AVK_E_HIP, "all-reduce of %llu sums: %s", (unsigned long long)n, hipGetErrorString(e));
    return AVK_E_OK;
}

int avk_tally_allreduce(avk_ctx *ctx, void *nccl_comm, uint64_t *tally) { return avk_counts_allreduce(ctx, nccl_comm, tally, AVK_TALLY_LEN); }

} /* extern "C" */
#endif
