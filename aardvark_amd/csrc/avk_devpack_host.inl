/*
 * avk_devpack_host.inl — the kernels around avk_devpack.inl and the host side of a device-packed batch.  Included by avk_host.hip.
 *
 * The host's part of avk_batch_upload / avk_compare_batch is now: queue the copies of the caller's arrays (straight from the caller's memory
 * when it is pinned — avk_host_alloc —, through a pinned bounce buffer filled by the host threads when it is not), queue a dozen
 * kernels, read ONE small state block back (sizes of the variable-length outputs, batch-level errors), queue the writers.  Device
 * buffers come from a pool the context keeps, so a call allocates nothing after the first one of its size.
 */

/* ---------------------------------------------------------------------------------- kernels */
namespace dpk = avk::dp;

__global__ void __launch_bounds__(256) avk_dp_expand_pairs_kernel(dpk::DpPairs c) { dpk::dp_expand_pairs(c, (uint64_t)blockIdx.x * 256u + threadIdx.x); }
__global__ void __launch_bounds__(256) avk_dp_merge_classify_kernel(dpk::DpMerge c) { dpk::dp_merge_classify(c, (uint64_t)blockIdx.x * 256u + threadIdx.x); }

__global__ void __launch_bounds__(256) avk_dp_widen_kernel(dpk::DpCompact c) { dpk::dp_widen(c, (uint64_t)blockIdx.x * 256u + threadIdx.x); }
__global__ void __launch_bounds__(256) avk_dp_widen_packed_kernel(dpk::DpPacked c) { dpk::dp_widen_packed(c, (uint64_t)blockIdx.x * 256u + threadIdx.x); }
__global__ void __launch_bounds__(256) avk_dp_widen_packed_multi_kernel(dpk::DpPackedMulti c) { dpk::dp_widen_packed_multi(c, (uint64_t)blockIdx.x * 256u + threadIdx.x); }
/* Exclusive prefix sum of a[i] + b[i] over n byte pairs (the packed form's offsets: calls per region, allele bytes per call): per-workgroup sums of 4096
 * elements, a one-workgroup scan of those sums, then every workgroup scans its own 4096 again from its base.  16 elements per thread, waves and workgroup
 * combined through LDS. */
#define AVK_PS_BLOCK 4096u
__device__ inline uint32_t avk_ps_thread_sum(const uint8_t *a, const uint8_t *b, uint64_t n, uint64_t i0, uint32_t (&v)[16]) {
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        const uint64_t i = i0 + k;
        v[k] = i < n ? (uint32_t)a[i] + (b ? (uint32_t)b[i] : 0u) : 0u; /* b == NULL: the sum of one array */
        s += v[k];
    }
    return s;
}
__global__ void __launch_bounds__(256) avk_ps_block_sums_kernel(const uint8_t *a, const uint8_t *b, uint64_t n, uint64_t *block_sums) {
    __shared__ uint32_t part[256];
    uint32_t v[16];
    part[threadIdx.x] = avk_ps_thread_sum(a, b, n, (uint64_t)blockIdx.x * AVK_PS_BLOCK + threadIdx.x * 16u, v);
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(1024) avk_ps_scan_sums_kernel(uint64_t *block_sums, uint32_t n_blocks, uint64_t *total) { /* one workgroup: in-place exclusive scan */
    __shared__ uint64_t part[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t x = i < n_blocks ? block_sums[i] : 0ull;
        part[threadIdx.x] = x;
        __syncthreads();
        for (uint32_t st = 1; st < 1024u; st <<= 1) { /* Hillis-Steele inclusive scan */
            const uint64_t y = threadIdx.x >= st ? part[threadIdx.x - st] : 0ull;
            __syncthreads();
            part[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 1023u) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry;
}
__global__ void __launch_bounds__(256) avk_ps_apply_kernel(const uint8_t *a, const uint8_t *b, uint64_t n, const uint64_t *block_base, uint64_t *out) {
    __shared__ uint32_t part[256];
    uint32_t v[16];
    const uint64_t i0 = (uint64_t)blockIdx.x * AVK_PS_BLOCK + threadIdx.x * 16u;
    const uint32_t mine = avk_ps_thread_sum(a, b, n, i0, v);
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t st = 1; st < 256u; st <<= 1) {
        const uint32_t y = threadIdx.x >= st ? part[threadIdx.x - st] : 0u;
        __syncthreads();
        part[threadIdx.x] += y;
        __syncthreads();
    }
    uint64_t run = block_base[blockIdx.x] + (part[threadIdx.x] - mine);
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        if (i0 + k < n) out[i0 + k] = run;
        run += v[k];
    }
}

__global__ void __launch_bounds__(256) avk_dp_variant_kernel(dpk::DpArgs a) { dpk::dp_variant(a, (uint64_t)blockIdx.x * 256u + threadIdx.x); }

/* dp_region for 256 regions; the workgroup's sums of the three scanned quantities and its count of regions per lane class go to
 * column-major block_sums[quantity][block]: no global atomics (56,000 waves adding to the one counter of the modal class took 1.3 ms of this kernel's 1.6) */
#define AVK_DP_NS 4 /* scanned quantities: per-call output words, blob words, sequence bytes, compact BASEPAIR groups */
#define AVK_DP_BS (AVK_DP_NS + AVK_FAST_CLASSES + avk::dp::DP_NEED_BUCKETS)
__global__ void __launch_bounds__(256) avk_dp_region_kernel(dpk::DpArgs a, uint64_t *block_sums) {
    __shared__ unsigned long long sums[AVK_DP_BS];
    if (threadIdx.x < AVK_DP_BS) sums[threadIdx.x] = 0;
    __syncthreads();
    uint32_t nc = 0, bw = 0, fc = 0, nb = 0xFFu, ng = 0;
    uint64_t sq = 0;
    dpk::dp_region(a, (uint64_t)blockIdx.x * 256u + threadIdx.x, nc, bw, sq, fc, nb, ng);
    if (__ballot(nb != 0xFFu)) /* class C regions: rare on a small-window genome */
        for (uint32_t c = 0; c < dpk::DP_NEED_BUCKETS; ++c) {
            const unsigned long long m = __ballot(nb == c);
            if (m && (threadIdx.x & 63u) == 0) atomicAdd(&sums[AVK_DP_NS + AVK_FAST_CLASSES + c], (unsigned long long)__popcll(m));
        }
    for (uint32_t c = 1; c <= AVK_FAST_CLASSES; ++c) { /* one LDS atomic per wave and class */
        const unsigned long long m = __ballot(fc == c);
        if (m && (threadIdx.x & 63u) == 0) atomicAdd(&sums[AVK_DP_NS - 1 + c], (unsigned long long)__popcll(m));
    }
    /* per-wave sums on the DPP network, 16 bits at a time so that 64 lanes cannot overflow a word; one LDS atomic per wave and quantity */
    const uint32_t nc_w = wv_sum_u32(nc); /* at most 64 x 60000 */
    const unsigned long long bw_w = (unsigned long long)wv_sum_u32(bw & 0xFFFFu) + ((unsigned long long)wv_sum_u32(bw >> 16) << 16);
    const unsigned long long sq_w = (unsigned long long)wv_sum_u32((uint32_t)sq & 0xFFFFu) + ((unsigned long long)wv_sum_u32((uint32_t)(sq >> 16) & 0xFFFFu) << 16) +
                                    ((unsigned long long)wv_sum_u32((uint32_t)(sq >> 32)) << 32);
    const uint32_t ng_w = wv_sum_u32(ng); /* at most 64 x 13 */
    if ((threadIdx.x & 63u) == 0) {
        if (ng_w) atomicAdd(&sums[3], (unsigned long long)ng_w);
        if (nc_w) atomicAdd(&sums[0], (unsigned long long)nc_w);
        if (bw_w) atomicAdd(&sums[1], bw_w);
        if (sq_w) atomicAdd(&sums[2], sq_w);
    }
    __syncthreads();
    if (threadIdx.x < AVK_DP_BS) block_sums[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sums[threadIdx.x]; /* one column per quantity */
}

/* one workgroup per column of the block sums: the scanned quantities get their exclusive scan in place and their total, the class and
 * workspace-bucket columns their sum (one workgroup walking all AVK_DP_BS columns of 14,000 rows took 0.2 ms of a 10 ms call) */
__global__ void __launch_bounds__(1024) avk_dp_scan_blocks_kernel(uint64_t *block_sums, uint32_t n_blocks, dpk::DpState *st) {
    __shared__ unsigned long long part[1024];
    const uint32_t q = blockIdx.x, t = threadIdx.x, per = (n_blocks + 1023u) / 1024u;
    uint64_t *col = block_sums + (size_t)q * n_blocks;
    const uint32_t lo = t * per < n_blocks ? t * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
    unsigned long long s = 0;
    for (uint32_t b = lo; b < hi; ++b) s += col[b];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const unsigned long long x = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    if (q < AVK_DP_NS) {
        unsigned long long run = part[t] - s;
        for (uint32_t b = lo; b < hi; ++b) {
            const unsigned long long v = col[b];
            col[b] = run;
            run += v;
        }
    }
    if (t == 1023) {
        const unsigned long long total = part[1023];
        if (q == 0) st->total_v = total;
        else if (q == 1) st->total_blob_words = total;
        else if (q == 2) st->total_seq = total;
        else if (q == 3) st->total_groups = total;
        else if (q < AVK_DP_NS + AVK_FAST_CLASSES) st->have[q - AVK_DP_NS] = total;
        else st->need_hist[q - AVK_DP_NS - AVK_FAST_CLASSES] = total;
    }
}

/* per-region offsets: the block's base + the exclusive scan inside the block */
__global__ void __launch_bounds__(256) avk_dp_scan_apply_kernel(dpk::DpArgs a, const uint64_t *block_sums) {
    __shared__ unsigned long long sh[AVK_DP_NS][256];
    const uint32_t t = threadIdx.x;
    const uint64_t r = (uint64_t)blockIdx.x * 256u + t;
    unsigned long long v[AVK_DP_NS] = {0, 0, 0, 0};
    if (r < a.in.n_regions) {
        const uint32_t tc = a.in.t_cnt_of(r), qc = a.in.q_cnt_of(r);
        const uint64_t toff = a.in.t_off_of(r), qoff = a.in.q_off_of(r), nv = a.in.n_variants;
        if (!(toff > nv || (uint64_t)tc > nv - toff || qoff > nv || (uint64_t)qc > nv - qoff)) { /* as dp_region counted it */
            v[0] = (unsigned long long)tc + qc;
            v[1] = a.rinfo[r].blob_bytes / 4u;
            v[2] = 5ull * a.rinfo[r].seq_stride;
            const uint32_t ps = a.rinfo[r].pre_status;
            v[3] = (ps & 0xFFFFu) ? 0u : 1u + (unsigned)__popc(ps >> 16); /* as dp_region counted the compact BASEPAIR groups */
        }
    }
    for (int q = 0; q < AVK_DP_NS; ++q) sh[q][t] = v[q];
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        unsigned long long x[AVK_DP_NS];
        for (int q = 0; q < AVK_DP_NS; ++q) x[q] = t >= d ? sh[q][t - d] : 0ull;
        __syncthreads();
        for (int q = 0; q < AVK_DP_NS; ++q) sh[q][t] += x[q];
        __syncthreads();
    }
    if (r < a.in.n_regions) {
        const uint64_t *bs = block_sums + blockIdx.x; /* column q of this workgroup: bs[q * gridDim.x] */
        const size_t col = gridDim.x;
        a.v_off[r] = (uint32_t)(bs[0] + sh[0][t] - v[0]);
        a.blob_off8[r] = (uint32_t)((bs[col] + sh[1][t] - v[1]) / 2ull);
        a.seq_off[r] = bs[2 * col] + sh[2][t] - v[2];
        a.bp_off[r] = (uint32_t)(bs[3 * col] + sh[3][t] - v[3]);
        if (r + 1 == a.in.n_regions) a.bp_off[r + 1] = (uint32_t)(bs[3 * col] + sh[3][t]);
    }
}

__global__ void avk_dp_lane_switch_kernel(dpk::DpArgs a) {
    if (blockIdx.x == 0 && threadIdx.x == 0) dpk::dp_lane_switch(a);
}

/* counting sort, pass 1: the histogram of the buckets (in LDS per workgroup, one global atomic per bucket the workgroup met) */
__global__ void __launch_bounds__(1024) avk_dp_hist_kernel(dpk::DpArgs a) {
    __shared__ uint32_t h[dpk::DP_NB];
    for (uint32_t k = threadIdx.x; k < dpk::DP_NB; k += 1024u) h[k] = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * 1024u + threadIdx.x;
    if (r < a.in.n_regions) {
        const uint32_t b = dpk::dp_bucket_of(a, r);
        a.bucket16[r] = (uint16_t)b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < dpk::DP_NB; k += 1024u)
        if (h[k]) atomicAdd(&a.st->hist[k], h[k]);
}

/* dp_bucket_bases by one workgroup: thread t scans its 9 consecutive buckets, the 256 partial sums are scanned in LDS (one thread walking the 2,304
 * buckets in global memory took 50 us) */
__global__ void __launch_bounds__(256) avk_dp_bucket_bases_kernel(dpk::DpArgs a) {
    static_assert(dpk::DP_NB % 256 == 0, "buckets per thread");
    enum { PER = dpk::DP_NB / 256 };
    __shared__ uint32_t part[256];
    dpk::DpState &s = *a.st;
    const uint32_t t = threadIdx.x;
    uint32_t h[PER], sum = 0;
    for (int k = 0; k < PER; ++k) h[k] = s.hist[t * PER + k], sum += h[k];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t x = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int k = 0; k < PER; ++k) {
        s.base[t * PER + k] = run;
        s.cursor[t * PER + k] = run;
        run += h[k];
    }
    if (t == 255) s.base[dpk::DP_NB] = run;
    __threadfence();
    __syncthreads();
    if (t == 0) dpk::dp_bucket_plan(a);
}

/* pass 2: a workgroup reserves its share of every bucket with one atomic and ranks its regions inside the share in LDS.  The order inside a
 * bucket is the caller's at the granularity of a workgroup (1024 regions) — a bucket holds regions of one cost, so nothing depends on it. */
__global__ void __launch_bounds__(1024) avk_dp_scatter_kernel(dpk::DpArgs a) {
    __shared__ uint32_t h[dpk::DP_NB];
    for (uint32_t k = threadIdx.x; k < dpk::DP_NB; k += 1024u) h[k] = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * 1024u + threadIdx.x;
    uint32_t b = 0;
    if (r < a.in.n_regions) {
        b = a.bucket16[r];
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < dpk::DP_NB; k += 1024u)
        if (h[k]) h[k] = atomicAdd(&a.st->cursor[k], h[k]);
    __syncthreads();
    if (r < a.in.n_regions) a.order[dpk::dp_order_slot(a, b, atomicAdd(&h[b], 1u))] = (uint32_t)r;
}

__global__ void __launch_bounds__(256) avk_dp_fast_records_kernel(dpk::DpArgs a, uint32_t n_tiles_total) {
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (tile >= n_tiles_total) return;
    uint32_t fc = 0;
    for (uint32_t c = 0; c < AVK_FAST_CLASSES; ++c)
        if (tile >= a.st->tile_first[c] && tile < a.st->tile_first[c] + a.st->fast_tiles[c]) fc = c;
    dpk::dp_fast_record(a, fc, tile - a.st->tile_first[fc], lane);
}

/* region records + blobs of the work-order positions [0, n_items): the regions the wave-per-region launches read from the start (classes C, B, bulk) — or
 * every region, when a run has no lane launches */
__global__ void __launch_bounds__(256) avk_dp_region_records_kernel(dpk::DpArgs a, uint32_t n_items) {
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k < n_items) dpk::dp_region_record(a, k);
}
__global__ void __launch_bounds__(256) avk_dp_region_records_wave_kernel(dpk::DpArgs a) {
    const uint32_t n_big = a.st->n_big, n_waves = gridDim.x * 4u;
    for (uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6); item < n_big; item += n_waves) dpk::dp_region_record_wave(a, item);
}

/* how many calls of the batch's range some region owns (DpIn::owned) */
__global__ void __launch_bounds__(256) avk_dp_count_owned_kernel(const uint8_t *owned, uint64_t n, dpk::DpState *st) {
    __shared__ uint32_t part;
    if (threadIdx.x == 0) part = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u, e = i + 16u < n ? i + 16u : n; i < e; ++i) mine += owned[i] ? 1u : 0u;
    if (mine) atomicAdd(&part, mine);
    __syncthreads();
    if (threadIdx.x == 0 && part) atomicAdd(&st->n_owned, part);
}

/* alt_ed of the calls dp_variant left to the host */
__global__ void avk_dp_patch_ed_kernel(dpk::DpVarInfo *vinfo, const uint32_t *idx, const uint32_t *ed, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        vinfo[idx[k]].alt_ed = ed[k];
        vinfo[idx[k]].flags &= ~(uint32_t)dpk::DP_VF_PENDING;
    }
}

__global__ void __launch_bounds__(256) avk_dp_unpack_kernel(dpk::DpOut o, uint8_t *pair_exact) {
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t word = 0, at = 0;
    const uint32_t need = dpk::dp_unpack_bp_need(o, r, word);
    if (o.bp_packed) { /* places in the spill list for the whole workgroup with one atomic: an exclusive scan of the lanes' needs */
        __shared__ uint32_t wave_sum[4], base;
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        uint32_t incl = need;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += y;
        }
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
            base = total ? atomicAdd(o.bp_spill_count, total) : 0u;
        }
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t k = 0; k < wv; ++k) before += wave_sum[k];
        at = base + before + incl - need;
    }
    dpk::dp_unpack(o, r, need, word, at);
    if (pair_exact && r < o.n_regions) pair_exact[r] = (o.region_out[4 * r] == 0 && o.region_out[4 * r + 1] != 0) ? 1 : 0; /* all_opt_haps[0].is_exact_match() */
}

/* ---------------------------------------------------------------------------------- host threads */
namespace {

/* A small persistent pool for the loops that are left on the host (copies between pageable and pinned memory): workers sleep on a condition
 * variable between loops; one loop at a time (calls on different contexts take turns). */
class AvkPool {
  public:
    static AvkPool &get() {
        static AvkPool p;
        return p;
    }
    /* runs fn(t) for t in [0, nt) on the workers (t = 0 on the caller) and returns when all are done */
    void run(unsigned nt, const std::function<void(unsigned)> &fn) {
        if (nt <= 1) {
            fn(0);
            return;
        }
        std::lock_guard<std::mutex> one(loop_mutex_);
        ensure(nt - 1);
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            want_ = nt - 1;
            next_ = 0;
            done_ = 0;
            gen_ += 1;
        }
        cv_.notify_all();
        fn(0);
        for (int spin = 0; spin < 4000 && done_.load(std::memory_order_acquire) != want_; ++spin) __builtin_ia32_pause();
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return done_ == want_; });
        fn_ = nullptr;
    }

  private:
    AvkPool() {}
    ~AvkPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            gen_ += 1;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void ensure(unsigned n) {
        while (workers_.size() < n) workers_.emplace_back([this] { work(); });
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *fn = nullptr;
            unsigned t = 0;
            for (int spin = 0; spin < 20000 && gen_.load(std::memory_order_acquire) == seen; ++spin) __builtin_ia32_pause();
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (gen_ != seen && next_ < want_); });
                if (stop_) return;
                t = ++next_; /* 1 .. want_ */
                seen = gen_; /* one index per loop and worker */
                fn = fn_;
            }
            (*fn)(t);
            {
                std::lock_guard<std::mutex> lk(m_);
                done_ += 1;
                if (done_ == want_) cv_done_.notify_all();
            }
        }
    }
    std::mutex m_, loop_mutex_;
    std::condition_variable cv_, cv_done_;
    std::vector<std::thread> workers_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned want_ = 0, next_ = 0;
    std::atomic<unsigned> done_{0};
    std::atomic<uint64_t> gen_{0};
    bool stop_ = false;
};

template <class F> void avk_parallel_for(uint64_t n, unsigned nt, F f) { /* f(thread, lo, hi) */
    if (nt <= 1 || n < 4096) {
        f(0u, (uint64_t)0, n);
        return;
    }
    AvkPool::get().run(nt, [&](unsigned t) { f(t, n * t / nt, n * (t + 1) / nt); });
}

unsigned avk_host_threads() {
    unsigned nt = avk_usable_cpus();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if (const char *e = getenv("AVK_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 256) nt = (unsigned)v;
    }
    return nt;
}

/* ---- pinned memory handed to the caller (avk_host_alloc): arrays that live there are copied by DMA without a host pass */
struct PinnedRange {
    const uint8_t *lo, *hi;
};
std::mutex g_pinned_mutex;
std::vector<PinnedRange> g_pinned;

bool is_pinned(const void *p, size_t bytes) {
    if (!p) return true;
    const uint8_t *lo = (const uint8_t *)p;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mutex);
        for (const PinnedRange &r : g_pinned)
            if (lo >= r.lo && lo + bytes <= r.hi) return true;
    }
    if (bytes < (1u << 20)) return false; /* small arrays: the bounce buffer costs less than asking the runtime */
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

} // namespace

/* ---- the context's pool of device buffers: every buffer is used in the order of the context's stream (side streams are forked from and joined
 * into it), so a released buffer may be handed out again at once — whatever is still queued on it runs before the next user's work */
static int pool_alloc(avk_ctx *ctx, void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mutex);
        int best = -1;
        for (size_t i = 0; i < ctx->pool.size(); ++i) {
            const auto &b = ctx->pool[i];
            if (b.used || b.bytes < bytes || b.bytes > 2 * bytes + (1u << 20)) continue;
            if (best < 0 || b.bytes < ctx->pool[(size_t)best].bytes) best = (int)i;
        }
        if (best >= 0) {
            ctx->pool[(size_t)best].used = true;
            ctx->pool_free_bytes -= ctx->pool[(size_t)best].bytes;
            *p = ctx->pool[(size_t)best].p;
            return 0;
        }
    }
    const bool timing = getenv("AVK_TIMING") != nullptr;
    const auto t_malloc = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, bytes);
    if (timing) fprintf(stderr, "avk pool: no cached buffer of %zu bytes, hipMalloc %.3f ms\n", bytes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_malloc).count());
    if (e == hipErrorOutOfMemory) { /* give the cached buffers back and try again */
        (void)hipGetLastError();
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(ctx->pool_mutex);
            for (size_t i = 0; i < ctx->pool.size();) {
                if (!ctx->pool[i].used) {
                    drop.push_back(ctx->pool[i].p);
                    ctx->pool_free_bytes -= ctx->pool[i].bytes;
                    ctx->pool.erase(ctx->pool.begin() + (long)i);
                } else
                    ++i;
            }
        }
        for (void *q : drop) (void)hipFree(q);
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(ctx, e == hipErrorOutOfMemory ? AVK_E_OOM : AVK_E_HIP, "device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(ctx->pool_mutex);
    ctx->pool.push_back({*p, bytes, true});
    return 0;
}

static void pool_release(avk_ctx *ctx, void *p) {
    if (!p) return;
    bool drop = false;
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mutex);
        for (size_t i = 0; i < ctx->pool.size(); ++i)
            if (ctx->pool[i].p == p) {
                if (ctx->pool_free_bytes + ctx->pool[i].bytes > (size_t)ctx->pool_cache_bytes) { /* beyond the cache limit: back to the runtime */
                    ctx->pool.erase(ctx->pool.begin() + (long)i);
                    drop = true;
                } else {
                    ctx->pool[i].used = false;
                    ctx->pool_free_bytes += ctx->pool[i].bytes;
                }
                break;
            }
    }
    if (drop) {
        const auto t_free = std::chrono::steady_clock::now();
        (void)hipFree(p);
        if (getenv("AVK_TIMING")) fprintf(stderr, "avk pool: cache limit reached, hipFree %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_free).count());
    }
}

/* avk_ctx_warmup: the buffers a device-packed batch of about n regions / nv calls takes from the pool, allocated, written once (fresh device memory is scrubbed
 * when it is first touched: 2.5 GB of them made the packing kernels of a process's first whole-genome call take 16 ms instead of 1.5) and put back.  A later
 * request takes a cached buffer that is at least as large and at most twice as large (pool_alloc): the sizes only have to be about right. */
static int pool_prewarm(avk_ctx *ctx, uint64_t n, uint64_t nv) {
    std::vector<size_t> sizes;
    auto add = [&](size_t bytes, int count) {
        for (int i = 0; i < count; ++i) sizes.push_back(bytes);
    };
    add((size_t)(n + 1) * 8, 5);   /* start, end, t_off, q_off, seq_off */
    add((size_t)(n + 1) * 4, 16);  /* counts, contig, per-region offsets, work order, the eight hand-over lists */
    add((size_t)(n + 1) * sizeof(dpk::DpRegionInfo), 1);
    add((size_t)(n + 1) * sizeof(AvkDevRegion), 1);
    add((size_t)n * 16 + 16, 2);   /* region records out, their caller-order form */
    add((size_t)n * 20 + 4, 1);
    add((size_t)(nv + 1) * 8, 3);  /* positions, allele offsets */
    add((size_t)(nv + 1) * 4, 5);  /* allele lengths, raw space, per-call words out and their caller-order form */
    add((size_t)nv + 16, 3);
    add((size_t)(nv + 1) * sizeof(dpk::DpVarInfo), 2); /* (and the call slots of the lane candidates) */
    add((size_t)n * 190, 1);       /* region blobs: 2.2 calls per region */
    add((size_t)n * 52, 1);        /* fast records */
    add((size_t)nv * 3 + 16, 1);   /* allele bytes */
    add((size_t)n * 10 + 16, 1);   /* the packed form's region and call records */
    add((size_t)nv * 5 + 16, 1);
    std::vector<void *> got;
    int rc = 0;
    for (size_t s : sizes) {
        void *p = nullptr;
        rc = pool_alloc(ctx, &p, s);
        if (rc) break;
        got.push_back(p);
        if (hipMemsetAsync(p, 0, s, ctx->stream) != hipSuccess) (void)hipGetLastError();
    }
    for (void *p : got) pool_release(ctx, p);
    return rc;
}

static void pool_destroy(avk_ctx *ctx) {
    for (auto &b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
    ctx->pool_free_bytes = 0;
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    ctx->h_bounce = nullptr;
    ctx->bounce_bytes = 0;
    if (ctx->h_dpstate) (void)hipHostFree(ctx->h_dpstate);
    ctx->h_dpstate = nullptr;
}

template <typename T> static int pool_alloc_t(avk_ctx *ctx, avk_dev_batch *db, T **p, size_t count) {
    void *q = nullptr;
    const int rc = pool_alloc(ctx, &q, count * sizeof(T));
    *p = (T *)q;
    if (!rc && db) db->pooled.push_back(q);
    return rc;
}

static int bounce_reserve(avk_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->bounce_bytes) return 0;
    bytes += bytes / 8 + (1u << 20);
    if (ctx->h_bounce) {
        AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipHostFree(ctx->h_bounce);
        ctx->h_bounce = nullptr;
        ctx->bounce_bytes = 0;
    }
    AVK_HIP(ctx, hipHostMalloc((void **)&ctx->h_bounce, bytes, hipHostMallocDefault));
    ctx->bounce_bytes = bytes;
    return 0;
}

/* ---- copies between the caller's arrays and the device ---------------------------------------------------------------------- */
struct CopySeg {
    const void *host; /* caller's array (source of an upload, destination of a download) */
    void *dev;
    size_t bytes;
    hipStream_t stream = nullptr;     /* uploads: a stream other than the context's for this array */
    hipEvent_t then_record = nullptr; /* uploads: recorded on that stream behind this array's copy */
};

/* Copies between PINNED host arrays and HBM by a kernel instead of a DMA engine (round 6).  hipMemcpyAsync hands such a copy to an SDMA engine, and which engine a
 * process gets is the driver's round robin: of the processes started one after the other on one box, every second or third copied at HALF the rate the first one got —
 * 23.5 instead of 47 GB/s in, 2.4 instead of 1.2 ms out, a whole-genome call 9.5 ms instead of 6.5, for the life of the process (profiles/r06_copy_engines.txt).  In a
 * synchronous call nothing else runs while its arrays cross the bus, so the compute units do it: every lane moves 16 bytes at a time straight from / to the pinned
 * pages (they are mapped into the device's address space), at the link's rate in every process.  The asynchronous boundary keeps the engines: its copies run beside
 * kernels. */
__global__ void __launch_bounds__(256) avk_copy_kernel(const uint8_t *src, uint8_t *dst, size_t bytes) {
    const size_t tid = (size_t)blockIdx.x * 256u + threadIdx.x, nthreads = (size_t)gridDim.x * 256u;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
        typedef unsigned int avk_v4u __attribute__((ext_vector_type(4)));
        const size_t n16 = bytes >> 4;
        const avk_v4u *s16 = (const avk_v4u *)src;
        avk_v4u *d16 = (avk_v4u *)dst;
        size_t i = tid;
        for (; i + 3 * nthreads < n16; i += 4 * nthreads) { /* four loads in flight per lane: the link's latency is microseconds */
            const avk_v4u a0 = __builtin_nontemporal_load(s16 + i), a1 = __builtin_nontemporal_load(s16 + i + nthreads), a2 = __builtin_nontemporal_load(s16 + i + 2 * nthreads),
                          a3 = __builtin_nontemporal_load(s16 + i + 3 * nthreads);
            __builtin_nontemporal_store(a0, d16 + i), __builtin_nontemporal_store(a1, d16 + i + nthreads), __builtin_nontemporal_store(a2, d16 + i + 2 * nthreads),
                __builtin_nontemporal_store(a3, d16 + i + 3 * nthreads);
        }
        for (; i < n16; i += nthreads) d16[i] = __builtin_nontemporal_load(s16 + i);
        for (size_t i = (n16 << 4) + tid; i < bytes; i += nthreads) dst[i] = src[i];
    } else if ((((uintptr_t)src | (uintptr_t)dst) & 3u) == 0) {
        const size_t n4 = bytes >> 2;
        const uint32_t *s4 = (const uint32_t *)src;
        uint32_t *d4 = (uint32_t *)dst;
        for (size_t i = tid; i < n4; i += nthreads) d4[i] = s4[i];
        for (size_t i = (n4 << 2) + tid; i < bytes; i += nthreads) dst[i] = src[i];
    } else
        for (size_t i = tid; i < bytes; i += nthreads) dst[i] = src[i];
}
/* Which of the two this context's synchronous copies use (option kernel_copies: 1 = decided by measurement, 0 = the engines, 2 = the kernel).  A probe copy of a
 * fresh buffer says nothing — it ran at 43-55 GB/s in processes whose calls then crawled — so the calls time THEMSELVES: two event records around the copies in of every
 * call that uses the engine (upload_device_packed); the first call whose arrays cross below 36 GB/s switches the context to the kernel for good.  A process that
 * drew a full-rate engine keeps it: it is 0.4 ms per whole-genome call faster than the kernel and leaves the CUs to dp_variant. */
static bool copies_by_kernel(const avk_ctx *ctx) { return ctx->kernel_copies == 2 || (ctx->kernel_copies == 1 && !ctx->engines_fast); }
static hipError_t kernel_copy(avk_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    if (!bytes) return hipSuccess;
    size_t blocks = (bytes / 16 + 255) / 256;
    const size_t cap = (size_t)ctx->n_cus * (size_t)(ctx->copy_blocks_per_cu > 0 ? ctx->copy_blocks_per_cu : 8);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(avk_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t *)src, (uint8_t *)dst, bytes);
    return hipGetLastError();
}

/* host -> device on the context's stream (or the segment's): straight from pinned arrays; pageable ones through the bounce buffer, piece by piece — the host threads
 * fill pieces while this thread queues the copy of every piece that is ready */
static int copy_in(avk_ctx *ctx, const std::vector<CopySeg> &segs) {
    struct Piece {
        const uint8_t *src;
        uint8_t *dev;
        size_t off, bytes;
        hipStream_t stream;
        hipEvent_t then_record; /* after this piece (the last of its segment) */
        bool direct;            /* pinned source: no staging */
    };
    std::vector<Piece> pieces;
    size_t staged = 0;
    const size_t piece_bytes = 4u << 20;
    for (const CopySeg &s : segs) {
        hipStream_t stream = s.stream ? s.stream : ctx->stream;
        if (!s.bytes || !s.host) {
            if (s.then_record) pieces.push_back({nullptr, nullptr, 0, 0, stream, s.then_record, true});
            continue;
        }
        if (is_pinned(s.host, s.bytes)) {
            pieces.push_back({(const uint8_t *)s.host, (uint8_t *)s.dev, 0, s.bytes, stream, s.then_record, true});
            continue;
        }
        for (size_t o = 0; o < s.bytes; o += piece_bytes) {
            const size_t nb = s.bytes - o < piece_bytes ? s.bytes - o : piece_bytes;
            pieces.push_back({(const uint8_t *)s.host + o, (uint8_t *)s.dev + o, staged, nb, stream, o + nb == s.bytes ? s.then_record : (hipEvent_t) nullptr, false});
            staged += (nb + 63) & ~(size_t)63;
        }
    }
    if (pieces.empty()) return 0;
    if (staged) {
        const int rc = bounce_reserve(ctx, staged);
        if (rc) return rc;
    }
    const bool timing = getenv("AVK_TIMING") != nullptr;
    size_t direct_bytes = 0;
    for (const Piece &p : pieces) direct_bytes += p.direct ? p.bytes : 0;
    const bool by_kernel = !ctx->up_stream && direct_bytes >= (8u << 20) && copies_by_kernel(ctx); /* (small batches: the engines' latency is what counts) */
    const bool timed_engine = !ctx->up_stream && !by_kernel && ctx->kernel_copies == 1 && direct_bytes >= (8u << 20) && staged == 0 && ctx->ev_cp0 && ctx->ev_cp1;
    if (timed_engine) (void)hipEventRecord(ctx->ev_cp0, ctx->stream);
    ctx->cp_timed_bytes = 0;
    auto issue = [&](const Piece &p) -> hipError_t { /* pieces are queued in segment order, whatever their source */
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        if (p.bytes && p.direct && by_kernel) e = kernel_copy(ctx, p.dev, p.src, p.bytes, p.stream); /* (a queued upload's copies run beside kernels: the engines) */
        else if (p.bytes) e = hipMemcpyAsync(p.dev, p.direct ? p.src : ctx->h_bounce + p.off, p.bytes, hipMemcpyHostToDevice, p.stream);
        const auto t1 = std::chrono::steady_clock::now();
        if (e == hipSuccess && p.then_record) e = hipEventRecord(p.then_record, p.stream);
        if (timing) {
            const double a = std::chrono::duration<double, std::milli>(t1 - t0).count(), b = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            if (a > 0.5 || b > 0.5)
                fprintf(stderr, "avk copy_in: queueing a copy of %zu bytes on the %s stream took %.3f ms, the event record behind it %.3f ms\n", p.bytes,
                        p.stream == ctx->stream ? "context's" : "side", a, b);
        }
        return e;
    };
    hipError_t herr = hipSuccess;
    const unsigned nt = avk_host_threads();
    if (!staged || nt <= 1) {
        for (size_t k = 0; k < pieces.size() && herr == hipSuccess; ++k) {
            if (!pieces[k].direct) memcpy(ctx->h_bounce + pieces[k].off, pieces[k].src, pieces[k].bytes);
            herr = issue(pieces[k]);
        }
    } else {
        std::atomic<size_t> next(0);
        std::vector<std::atomic<uint8_t>> ready(pieces.size());
        for (size_t k = 0; k < pieces.size(); ++k) ready[k].store(pieces[k].direct ? 1 : 0);
        AvkPool::get().run(nt + 1, [&](unsigned t) {
            if (t == 0) { /* this thread queues, the others fill */
                for (size_t k = 0; k < pieces.size() && herr == hipSuccess; ++k) {
                    while (!ready[k].load(std::memory_order_acquire)) std::this_thread::yield();
                    herr = issue(pieces[k]);
                }
                return;
            }
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= pieces.size()) break;
                if (pieces[k].direct) continue;
                memcpy(ctx->h_bounce + pieces[k].off, pieces[k].src, pieces[k].bytes);
                ready[k].store(1, std::memory_order_release);
            }
        });
    }
    if (herr != hipSuccess) return fail(ctx, AVK_E_HIP, "host to device copy failed: %s", hipGetErrorString(herr));
    if (timed_engine && hipEventRecord(ctx->ev_cp1, ctx->stream) == hipSuccess) ctx->cp_timed_bytes = direct_bytes; /* (read by engine_rate_check once the stream has been waited for) */
    return 0;
}
/* behind a host synchronisation with the context's stream: how fast the engine moved the arrays of the copy_in before it */
static void engine_rate_check(avk_ctx *ctx) {
    if (!ctx->cp_timed_bytes) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_cp0, ctx->ev_cp1) == hipSuccess && ms > 0) {
        ctx->engine_in_gbs = ctx->cp_timed_bytes / (ms * 1e-3) / 1e9;
        if (ms > ctx->cp_timed_bytes / 36e9 * 1e3 + 0.15) { /* below 36 GB/s, with 0.15 ms for the dozen copies' own latencies (a rank's shard is 10 MB) */
            ctx->engines_fast = false;
            if (getenv("AVK_TIMING")) fprintf(stderr, "avk copies: the DMA engine moved this call's arrays at %.1f GB/s: synchronous calls copy by kernel from now on\n", ctx->engine_in_gbs);
        }
    } else
        (void)hipGetLastError();
    ctx->cp_timed_bytes = 0;
}

/* device -> host: queues the copies (pinned destinations directly, the others into the bounce buffer); finish_copy_out waits for the stream and
 * moves the bounced parts into the caller's arrays on the host threads */
struct CopyOut {
    struct Part {
        void *host;
        size_t off, bytes;
    };
    std::vector<Part> parts;
};
static int copy_out(avk_ctx *ctx, const std::vector<CopySeg> &segs, CopyOut *co) {
    size_t staged = 0;
    for (const CopySeg &s : segs)
        if (s.bytes && s.host && !is_pinned(s.host, s.bytes)) staged += (s.bytes + 63) & ~(size_t)63;
    if (staged) {
        const int rc = bounce_reserve(ctx, staged);
        if (rc) return rc;
    }
    size_t off = 0, pinned_bytes = 0;
    for (const CopySeg &s : segs) pinned_bytes += s.bytes && s.host && is_pinned(s.host, s.bytes) ? s.bytes : 0;
    const bool by_kernel = pinned_bytes >= (8u << 20) && copies_by_kernel(ctx);
    for (const CopySeg &s : segs) {
        if (!s.bytes || !s.host) continue;
        if (is_pinned(s.host, s.bytes)) {
            if (by_kernel) AVK_HIP(ctx, kernel_copy(ctx, (void *)s.host, s.dev, s.bytes, ctx->stream));
            else AVK_HIP(ctx, hipMemcpyAsync((void *)s.host, s.dev, s.bytes, hipMemcpyDeviceToHost, ctx->stream));
            continue;
        }
        AVK_HIP(ctx, hipMemcpyAsync(ctx->h_bounce + off, s.dev, s.bytes, hipMemcpyDeviceToHost, ctx->stream));
        co->parts.push_back({(void *)s.host, off, s.bytes});
        off += (s.bytes + 63) & ~(size_t)63;
    }
    return 0;
}
static int finish_copy_out(avk_ctx *ctx, const CopyOut &co) {
    AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (co.parts.empty()) return 0;
    struct Piece {
        uint8_t *dst;
        const uint8_t *src;
        size_t bytes;
    };
    std::vector<Piece> pieces;
    const size_t piece_bytes = 4u << 20;
    for (const auto &p : co.parts)
        for (size_t o = 0; o < p.bytes; o += piece_bytes) pieces.push_back({(uint8_t *)p.host + o, ctx->h_bounce + p.off + o, p.bytes - o < piece_bytes ? p.bytes - o : piece_bytes});
    std::atomic<size_t> next(0);
    const unsigned nt = pieces.size() > 1 ? avk_host_threads() : 1;
    AvkPool::get().run(nt, [&](unsigned) {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= pieces.size()) break;
            memcpy(pieces[k].dst, pieces[k].src, pieces[k].bytes);
        }
    });
    return 0;
}

/* ---- upload ---------------------------------------------------------------------------------------------------------------- */
static void release_pooled(avk_ctx *ctx, avk_dev_batch *db) {
    for (void *p : db->pooled) pool_release(ctx, p);
    db->pooled.clear();
}

/* the arrays of an avk_packed_batch already in device memory (a staging slot of avk_compare_packed_submit, filled on the context's copy stream): nothing is copied,
 * the context's stream waits for `ready`; the block stays with the caller (the allele bytes and raw spaces are read until the batch's solve is over) */
struct PackedOnDevice {
    uint8_t *t_cnt, *q_cnt, *a0_len, *a1_len, *var_type_zyg, *alleles;
    uint32_t *start, *raw;
    uint16_t *len, *contig, *rel_pos;
    hipEvent_t ready;
};

/* b: the batch in the wide form, or NULL and cb: the batch in the compact form (avk_compact_batch: half the bytes over PCIe, widened on the device) */
/* mb: a batch of MultiRegions (the merge path): one region per input pair is made on the device (dp_expand_pairs) */
/* (pm: the packed form of a multi batch; `mb` then only carries n_regions, n_inputs, n_variants, allele_bytes and allele_bytes_len) */
static int upload_device_packed(avk_ctx *ctx, const avk_region_batch *b, const avk_compact_batch *cb, bool pairs_mode, avk_dev_batch **out, const avk_multi_batch *mb = nullptr,
                                const avk_packed_batch *pk = nullptr, const avk_packed_multi_batch *pm = nullptr, const PackedOnDevice *pre = nullptr) {
    const uint32_t mk = mb ? mb->n_inputs : 0, mppr = mk * (mk - (mk ? 1u : 0u)) / 2;
    const uint64_t n = pk ? pk->n_regions : (b ? b->n_regions : (cb ? cb->n_regions : mb->n_regions * mppr)), nv = pk ? pk->n_variants : (b ? b->n_variants : (cb ? cb->n_variants : mb->n_variants)),
                   alen = pk ? pk->allele_bytes_len : (b ? b->allele_bytes_len : (cb ? cb->allele_bytes_len : mb->allele_bytes_len));
    if (pk && n && (!pk->start || !pk->len || !pk->t_cnt || !pk->q_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (pk && nv && (!pk->var_rel_pos || !pk->var_type_zyg || !pk->a0_len || !pk->a1_len || !pk->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    if (pm && pm->n_regions && (!pm->start || !pm->len || !pm->in_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (pm && nv && (!pm->var_rel_pos || !pm->var_type_zyg || !pm->a0_len || !pm->a1_len || !pm->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    if (mb && !pm && mb->n_regions && (!mb->start || !mb->end || !mb->in_off || !mb->in_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (mb && !pm && nv && (!mb->var_pos || !mb->var_type || !mb->var_zyg || !mb->a0_off || !mb->a0_len || !mb->a1_off || !mb->a1_len || !mb->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    if (n > 0x7FFFFFFFull || nv > 0x7FFFFFFFull) return fail(ctx, AVK_E_ARG, "batch too large (more than 2^31 regions or variants); split it");
    if (b && n && (!b->start || !b->end || !b->t_off || !b->t_cnt || !b->q_off || !b->q_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (b && nv && (!b->var_pos || !b->var_type || !b->var_zyg || !b->a0_off || !b->a0_len || !b->a1_off || !b->a1_len || !b->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    if (cb && n && (!cb->start || !cb->len || !cb->v_off || !cb->t_cnt || !cb->q_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (cb && nv && (!cb->var_pos || !cb->var_type_zyg || !cb->a_off || !cb->a0_len || !cb->a1_len || !cb->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    if (cb && alen > 0xFFFFFFFFull) return fail(ctx, AVK_E_ARG, "the compact form holds at most 2^32 allele bytes");
    const bool has_contig = pm ? pm->contig_idx != nullptr : (pk ? pk->contig_idx != nullptr : (b ? b->contig_idx != nullptr : (cb ? cb->contig_idx != nullptr : mb->contig_idx != nullptr))),
               has_raw = pm ? pm->var_raw_space != nullptr : (pk ? pk->var_raw_space != nullptr : (b ? b->var_raw_space != nullptr : (cb ? cb->var_raw_space != nullptr : mb->var_raw_space != nullptr)));
    const uint8_t *host_alleles = pk ? pk->allele_bytes : (b ? b->allele_bytes : (cb ? cb->allele_bytes : mb->allele_bytes));
    const bool timing = getenv("AVK_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
    const auto t_start = now();
    hipStream_t s = ctx->up_stream ? ctx->up_stream : ctx->stream; /* (the asynchronous boundary packs on a stream of its own, beside the solve of the batch before) */
    uint64_t *pk_totals = nullptr; /* packed form: device words {sum of the call counts, sum of the allele lengths} */
    avk_dev_batch *db = new avk_dev_batch();
    db->dev_packed = true;
    db->n_regions = n;
    db->n_variants_host = nv;
    std::vector<void *> temps; /* released (to the pool) at the end of the upload */
    int rc = 0;
    auto tmp = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (!rc) rc = pool_alloc(ctx, &p, bytes);
        if (p) temps.push_back(p);
        return p;
    };
    auto kept = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (!rc) rc = pool_alloc(ctx, &p, bytes);
        if (p) db->pooled.push_back(p);
        return p;
    };
    auto bail = [&](int code) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize((ctx->up_side ? ctx->up_side : ctx->lane_stream4)); /* (the side stream of the upload: prefix sums, widening, the cleared scratch) */
        for (void *p : temps) pool_release(ctx, p);
        release_pooled(ctx, db);
        delete db;
        return code;
    };
    /* the caller's arrays in HBM; the four that dp_unpack reads after the solve stay with the batch */
    dpk::DpArgs a;
    memset(&a, 0, sizeof(a));
    /* (the inputs and the packer's intermediates stay with the batch: region records of lane regions are written when a launch needs them) */
    /* a compare batch in the packed form is read from the packed arrays themselves (DpIn::pk_*, round 6): no wide arrays, no widening pass */
    const bool packed_src = pk != nullptr && ctx->packed_source;
    bool early_variant = false, variant_done = false; /* dp_variant queued on the side stream of a packed upload, under its copies */
    auto tmp_or_kept = [&](size_t bytes) -> void * { return kept(bytes); };
    auto wide = [&](size_t bytes) -> void * { return packed_src ? nullptr : kept(bytes); };
    uint64_t *d_start = (uint64_t *)wide((n + 1) * 8), *d_end = (uint64_t *)wide((n + 1) * 8);
    db->d_in_t_off = (uint64_t *)wide((n + 1) * 8), db->d_in_q_off = (uint64_t *)wide((n + 1) * 8);
    db->d_in_t_cnt = (uint32_t *)wide((n + 1) * 4), db->d_in_q_cnt = (uint32_t *)wide((n + 1) * 4);
    uint32_t *d_contig = has_contig ? (uint32_t *)wide((n + 1) * 4) : nullptr;
    uint64_t *d_pos = (uint64_t *)wide((nv + 1) * 8), *d_a0o = (uint64_t *)wide((nv + 1) * 8), *d_a1o = (uint64_t *)wide((nv + 1) * 8);
    uint32_t *d_a0l = (uint32_t *)wide((nv + 1) * 4), *d_a1l = (uint32_t *)wide((nv + 1) * 4);
    uint32_t *d_raw = has_raw ? (pre ? pre->raw : (uint32_t *)tmp_or_kept((nv + 1) * 4)) : nullptr;
    uint8_t *d_type = (uint8_t *)wide(nv + 16), *d_zyg = (uint8_t *)wide(nv + 16), *d_alleles = pre ? pre->alleles : (uint8_t *)tmp_or_kept(alen + 16);
    /* intermediates */
    const uint32_t n_blocks = (uint32_t)((n + 255) / 256);
    a.vinfo = (dpk::DpVarInfo *)kept((nv + 1) * sizeof(dpk::DpVarInfo));
    a.rinfo = (dpk::DpRegionInfo *)kept((n + 1) * sizeof(dpk::DpRegionInfo));
    a.st = (dpk::DpState *)kept(sizeof(dpk::DpState));
    a.pending = (uint32_t *)tmp((nv + 1) * 4);
    a.slots = (dpk::DpSlot *)tmp((nv + 1) * sizeof(dpk::DpSlot));
    a.bucket16 = (uint16_t *)tmp((n + 16) * 2);
    db->d_voff = (uint32_t *)kept((n + 1) * 4);
    a.v_off = db->d_voff;
    a.blob_off8 = (uint32_t *)kept((n + 1) * 4);
    a.seq_off = (uint64_t *)kept((n + 1) * 8);
    db->d_bp_off = (uint32_t *)kept((n + 2) * 4);
    a.bp_off = db->d_bp_off;
    if (db->d_bp_off) (void)hipMemsetAsync(db->d_bp_off, 0, 8, s); /* (an empty batch: bp_off[0] = 0) */
    a.order = (uint32_t *)kept((n + 1) * 4);
    a.big_list = (uint32_t *)kept((n + 1) * 4);
    uint64_t *d_block_sums = (uint64_t *)tmp(((size_t)n_blocks + 1) * AVK_DP_BS * 8);
    if (!ctx->h_dpstate && !rc) {
        hipError_t e = hipHostMalloc((void **)&ctx->h_dpstate, sizeof(dpk::DpState) + 16, hipHostMallocDefault);
        if (e != hipSuccess) rc = fail(ctx, AVK_E_HIP, "pinned state block: %s", hipGetErrorString(e));
    }
    if (!ctx->d_contig_tab && !rc) rc = fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    if (rc) return bail(rc);
    const auto t_alloc = now();
    auto mark = [&](int k) { /* AVK_TIMING: a mark of the call's device timeline on the context's stream */
        if (!timing) return;
        if (!ctx->ev_tl[k] && hipEventCreate(&ctx->ev_tl[k]) != hipSuccess) {
            ctx->ev_tl[k] = nullptr;
            (void)hipGetLastError();
            return;
        }
        (void)hipEventRecord(ctx->ev_tl[k], s);
    };
    mark(0);
    if (pk) { /* the packed arrays as they are, two prefix sums for the offsets they leave out, one kernel that writes the wide arrays */
        /* (the packed arrays stay with the batch when they are what the packer and the later record writers read; a staging slot's stay with its ticket) */
        auto src = [&](size_t bytes) -> void * { return packed_src ? kept(bytes) : tmp(bytes); };
        uint16_t *p_contig = has_contig ? (pre ? pre->contig : (uint16_t *)src((n + 1) * 2)) : nullptr, *p_len = pre ? pre->len : (uint16_t *)src((n + 1) * 2),
                 *p_rel = pre ? pre->rel_pos : (uint16_t *)src((nv + 1) * 2);
        uint32_t *p_start = pre ? pre->start : (uint32_t *)src((n + 1) * 4);
        uint8_t *p_tc = pre ? pre->t_cnt : (uint8_t *)src(n + 16), *p_qc = pre ? pre->q_cnt : (uint8_t *)src(n + 16), *p_tz = pre ? pre->var_type_zyg : (uint8_t *)src(nv + 16),
                *p_a0 = pre ? pre->a0_len : (uint8_t *)src(nv + 16), *p_a1 = pre ? pre->a1_len : (uint8_t *)src(nv + 16);
        const uint32_t nb_r = (uint32_t)((n + AVK_PS_BLOCK - 1) / AVK_PS_BLOCK), nb_v = (uint32_t)((nv + AVK_PS_BLOCK - 1) / AVK_PS_BLOCK);
        uint64_t *p_voff = (uint64_t *)src((n + 1) * 8), *p_aoff = (uint64_t *)src((nv + 1) * 8), *p_sums = (uint64_t *)tmp(((size_t)nb_r + nb_v + 4) * 8);
        if (rc) return bail(rc);
        /* All copies on the context's stream, counts and lengths first; the two prefix sums (which need nothing else) and the widening kernel (everything but the
         * allele bytes) run on a stream of their own beside the copies that follow: dp_variant, the first kernel that needs every byte, starts 0.24 ms earlier.
         * (The copies themselves stay on ONE stream: a large copy queued on a second stream now and then blocks the host for 7 to 9 ms inside hipMemcpyAsync —
         * the runtime bringing up another copy engine — which made one call in fifty 16 ms long.) */
        hipStream_t side = (ctx->up_side ? ctx->up_side : ctx->lane_stream4);
        auto side_fail = [&](int code) { /* nothing of this call may still be in flight on the side stream when its buffers go back */
            (void)hipStreamSynchronize(side);
            return bail(code);
        };
        if (pre) { /* the arrays are in a staging slot already (or on their way there on the copy stream) */
            hipError_t ep = hipStreamWaitEvent(s, pre->ready, 0);
            if (ep == hipSuccess) ep = hipEventRecord(ctx->ev_copy_fork, s);
            if (ep == hipSuccess) ep = hipEventRecord(ctx->ev_copy_mid, s);
            if (ep == hipSuccess) ep = hipEventRecord(ctx->ev_copy_alleles, s);
            if (ep != hipSuccess) rc = fail(ctx, AVK_E_HIP, "packed upload: %s", hipGetErrorString(ep));
        } else
        /* (round 6: the allele bytes cross right behind the lengths — dp_variant needs nothing else, and runs on the side stream under the copies that follow) */
        rc = copy_in(ctx, {{pk->t_cnt, p_tc, n}, {pk->q_cnt, p_qc, n}, {pk->a0_len, p_a0, nv}, {pk->a1_len, p_a1, nv, nullptr, ctx->ev_copy_fork},
                           {pk->allele_bytes, d_alleles, alen, nullptr, ctx->ev_copy_alleles}, {pk->start, p_start, n * 4},
                           {pk->len, p_len, n * 2}, {pk->contig_idx, p_contig, has_contig ? n * 2 : 0}, {pk->var_rel_pos, p_rel, nv * 2}, {pk->var_type_zyg, p_tz, nv, nullptr, ctx->ev_copy_mid},
                           {pk->var_raw_space, d_raw, has_raw ? nv * 4 : 0}});
        early_variant = packed_src && nv != 0;
        mark(1);
        if (rc) return bail(rc);
        hipError_t ec = hipStreamWaitEvent(side, ctx->ev_copy_fork, 0); /* (also orders the side stream behind everything queued on the context's stream before) */
        if (ec != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "packed upload: %s", hipGetErrorString(ec)));
        if (n) {
            hipLaunchKernelGGL(avk_ps_block_sums_kernel, dim3(nb_r), dim3(256), 0, side, (const uint8_t *)p_tc, (const uint8_t *)p_qc, n, p_sums);
            hipLaunchKernelGGL(avk_ps_scan_sums_kernel, dim3(1), dim3(1024), 0, side, p_sums, nb_r, p_sums + nb_r + nb_v);
            hipLaunchKernelGGL(avk_ps_apply_kernel, dim3(nb_r), dim3(256), 0, side, (const uint8_t *)p_tc, (const uint8_t *)p_qc, n, (const uint64_t *)p_sums, p_voff);
        }
        if (nv) {
            hipLaunchKernelGGL(avk_ps_block_sums_kernel, dim3(nb_v), dim3(256), 0, side, (const uint8_t *)p_a0, (const uint8_t *)p_a1, nv, p_sums + nb_r);
            hipLaunchKernelGGL(avk_ps_scan_sums_kernel, dim3(1), dim3(1024), 0, side, p_sums + nb_r, nb_v, p_sums + nb_r + nb_v + 1);
            hipLaunchKernelGGL(avk_ps_apply_kernel, dim3(nb_v), dim3(256), 0, side, (const uint8_t *)p_a0, (const uint8_t *)p_a1, nv, (const uint64_t *)(p_sums + nb_r), p_aoff);
        }
        /* the totals must be what the caller said: n_variants calls, allele_bytes_len bytes (two words back, with the packer's state block) */
        pk_totals = p_sums + nb_r + nb_v;
        dpk::DpPacked c;
        memset(&c, 0, sizeof(c));
        c.contig_idx = p_contig, c.len = p_len, c.rel_pos = p_rel, c.start = p_start, c.var_raw = d_raw, c.t_cnt = p_tc, c.q_cnt = p_qc, c.var_type_zyg = p_tz, c.a0_len = p_a0, c.a1_len = p_a1,
        c.v_off = p_voff, c.a_off = p_aoff, c.n_regions = n, c.n_variants = nv;
        c.w_contig = d_contig, c.w_t_cnt = db->d_in_t_cnt, c.w_q_cnt = db->d_in_q_cnt, c.w_a0_len = d_a0l, c.w_a1_len = d_a1l, c.w_raw = nullptr, c.w_start = d_start, c.w_end = d_end,
        c.w_t_off = db->d_in_t_off, c.w_q_off = db->d_in_q_off, c.w_pos = d_pos, c.w_a0_off = d_a0o, c.w_a1_off = d_a1o, c.w_type = d_type, c.w_zyg = d_zyg;
        const uint64_t m = n > nv ? n : nv;
        hipError_t ew = hipSuccess;
        if (early_variant) {
            /* Variant::alt_ed for every call, on the side stream behind the prefix sums and the allele bytes, while the region arrays still cross the bus: the state
             * block is cleared here (dp_variant counts the calls it leaves to the host); the context's stream waits for ev_copy_join below as before */
            a.in.pk_a0 = p_a0, a.in.pk_a1 = p_a1, a.in.pk_aoff = p_aoff, a.in.pk_start = p_start; /* (what dp_variant reads of the packed source; the rest follows below) */
            a.in.alleles = d_alleles, a.in.n_variants = nv, a.in.alleles_len = alen, a.in.n_regions = n;
            ew = hipMemsetAsync(a.st, 0, sizeof(dpk::DpState), side);
            if (ew == hipSuccess) ew = hipStreamWaitEvent(side, ctx->ev_copy_alleles, 0);
            if (ew == hipSuccess) {
                hipLaunchKernelGGL(avk_dp_variant_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, side, a);
                ew = hipGetLastError();
            }
            variant_done = ew == hipSuccess;
        }
        if (ew == hipSuccess) ew = hipStreamWaitEvent(side, ctx->ev_copy_mid, 0);
        if (packed_src) { /* the packer reads these as they are */
            a.in.pk_start = p_start, a.in.pk_len = p_len, a.in.pk_contig = p_contig, a.in.pk_rel = p_rel, a.in.pk_tc = p_tc, a.in.pk_qc = p_qc, a.in.pk_tz = p_tz, a.in.pk_a0 = p_a0,
            a.in.pk_a1 = p_a1, a.in.pk_voff = p_voff, a.in.pk_aoff = p_aoff;
            db->d_pk_voff = p_voff, db->d_pk_tc = p_tc, db->d_pk_qc = p_qc;
        } else if (m && ew == hipSuccess) {
            hipLaunchKernelGGL(avk_dp_widen_packed_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, side, c);
            ew = hipGetLastError();
        }
        if (ew == hipSuccess) ew = hipEventRecord(ctx->ev_copy_join, side);
        if (ew == hipSuccess) ew = hipStreamWaitEvent(s, ctx->ev_copy_join, 0); /* from here on the context's stream has the wide arrays (and, being behind its own copies, the allele bytes) */
        if (ew != hipSuccess) return side_fail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(ew)));
    } else if (b) {
        std::vector<CopySeg> segs = {
            {b->start, d_start, n * 8}, {b->end, d_end, n * 8}, {b->t_off, db->d_in_t_off, n * 8}, {b->q_off, db->d_in_q_off, n * 8},
            {b->t_cnt, db->d_in_t_cnt, n * 4}, {b->q_cnt, db->d_in_q_cnt, n * 4}, {b->contig_idx, d_contig, b->contig_idx ? n * 4 : 0},
            {b->var_pos, d_pos, nv * 8}, {b->a0_off, d_a0o, nv * 8}, {b->a1_off, d_a1o, nv * 8}, {b->a0_len, d_a0l, nv * 4}, {b->a1_len, d_a1l, nv * 4},
            {b->var_raw_space, d_raw, b->var_raw_space ? nv * 4 : 0}, {b->var_type, d_type, nv}, {b->var_zyg, d_zyg, nv}, {b->allele_bytes, d_alleles, alen}};
        rc = copy_in(ctx, segs);
        if (rc) return bail(rc);
    } else if (mb) { /* the MultiRegions as they are, one kernel that writes a region per input pair; the call arrays are shared by all pairs */
        const uint64_t nm = mb->n_regions;
        uint32_t *m_contig = has_contig ? (uint32_t *)tmp((nm + 1) * 4) : nullptr;
        uint64_t *m_start = (uint64_t *)tmp((nm + 1) * 8), *m_end = (uint64_t *)tmp((nm + 1) * 8);
        db->d_m_in_off = (uint64_t *)kept((nm * mk + 1) * 8);
        db->d_m_in_cnt = (uint32_t *)kept((nm * mk + 1) * 4);
        db->d_in_zyg = d_zyg;
        db->n_multi = nm, db->m_inputs = mk;
        if (rc) return bail(rc);
        if (pm) { /* the packed arrays as they are; in_off and a_off by prefix sums; one kernel writes the wide MultiRegion arrays the pair expansion reads */
            const uint64_t ni = nm * mk;
            uint16_t *p_contig = has_contig ? (uint16_t *)tmp((nm + 1) * 2) : nullptr, *p_len = (uint16_t *)tmp((nm + 1) * 2), *p_rel = (uint16_t *)tmp((nv + 1) * 2);
            uint32_t *p_start = (uint32_t *)tmp((nm + 1) * 4);
            uint8_t *p_ic = (uint8_t *)tmp(ni + 16), *p_tz = (uint8_t *)tmp(nv + 16), *p_a0 = (uint8_t *)tmp(nv + 16), *p_a1 = (uint8_t *)tmp(nv + 16);
            const uint32_t nb_r = (uint32_t)((ni + AVK_PS_BLOCK - 1) / AVK_PS_BLOCK), nb_v = (uint32_t)((nv + AVK_PS_BLOCK - 1) / AVK_PS_BLOCK);
            uint64_t *p_aoff = (uint64_t *)tmp((nv + 1) * 8), *p_sums = (uint64_t *)tmp(((size_t)nb_r + nb_v + 4) * 8);
            if (rc) return bail(rc);
            hipStream_t side = (ctx->up_side ? ctx->up_side : ctx->lane_stream4); /* as for avk_packed_batch: copies on the context's stream, counts first; prefix sums and widening beside them */
            auto side_fail = [&](int code) {
                (void)hipStreamSynchronize(side);
                return bail(code);
            };
            rc = copy_in(ctx, {{pm->in_cnt, p_ic, ni}, {pm->a0_len, p_a0, nv}, {pm->a1_len, p_a1, nv, nullptr, ctx->ev_copy_fork}, {pm->start, p_start, nm * 4}, {pm->len, p_len, nm * 2},
                               {pm->contig_idx, p_contig, has_contig ? nm * 2 : 0}, {pm->var_rel_pos, p_rel, nv * 2}, {pm->var_type_zyg, p_tz, nv, nullptr, ctx->ev_copy_mid},
                               {pm->var_raw_space, d_raw, has_raw ? nv * 4 : 0}, {pm->allele_bytes, d_alleles, alen}});
            if (rc) return bail(rc);
            hipError_t ec = hipStreamWaitEvent(side, ctx->ev_copy_fork, 0);
            if (ec != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "packed upload: %s", hipGetErrorString(ec)));
            if (ni) {
                hipLaunchKernelGGL(avk_ps_block_sums_kernel, dim3(nb_r), dim3(256), 0, side, (const uint8_t *)p_ic, (const uint8_t *)nullptr, ni, p_sums);
                hipLaunchKernelGGL(avk_ps_scan_sums_kernel, dim3(1), dim3(1024), 0, side, p_sums, nb_r, p_sums + nb_r + nb_v);
                hipLaunchKernelGGL(avk_ps_apply_kernel, dim3(nb_r), dim3(256), 0, side, (const uint8_t *)p_ic, (const uint8_t *)nullptr, ni, (const uint64_t *)p_sums, db->d_m_in_off);
            }
            if (nv) {
                hipLaunchKernelGGL(avk_ps_block_sums_kernel, dim3(nb_v), dim3(256), 0, side, (const uint8_t *)p_a0, (const uint8_t *)p_a1, nv, p_sums + nb_r);
                hipLaunchKernelGGL(avk_ps_scan_sums_kernel, dim3(1), dim3(1024), 0, side, p_sums + nb_r, nb_v, p_sums + nb_r + nb_v + 1);
                hipLaunchKernelGGL(avk_ps_apply_kernel, dim3(nb_v), dim3(256), 0, side, (const uint8_t *)p_a0, (const uint8_t *)p_a1, nv, (const uint64_t *)(p_sums + nb_r), p_aoff);
            }
            pk_totals = p_sums + nb_r + nb_v;
            dpk::DpPackedMulti w;
            memset(&w, 0, sizeof(w));
            w.contig_idx = p_contig, w.len = p_len, w.rel_pos = p_rel, w.start = p_start, w.var_raw = d_raw, w.in_cnt = p_ic, w.var_type_zyg = p_tz, w.a0_len = p_a0, w.a1_len = p_a1,
            w.in_off = db->d_m_in_off, w.a_off = p_aoff, w.n_multi = nm, w.n_variants = nv, w.k = mk;
            w.w_contig = m_contig, w.w_in_cnt = db->d_m_in_cnt, w.w_a0_len = d_a0l, w.w_a1_len = d_a1l, w.w_raw = nullptr, w.w_start = m_start, w.w_end = m_end, w.w_pos = d_pos,
            w.w_a0_off = d_a0o, w.w_a1_off = d_a1o, w.w_type = d_type, w.w_zyg = d_zyg;
            const uint64_t mx = nm > nv ? nm : nv;
            hipError_t ew = hipStreamWaitEvent(side, ctx->ev_copy_mid, 0);
            if (mx && ew == hipSuccess) {
                hipLaunchKernelGGL(avk_dp_widen_packed_multi_kernel, dim3((unsigned)((mx + 255) / 256)), dim3(256), 0, side, w);
                ew = hipGetLastError();
            }
            if (ew == hipSuccess) ew = hipEventRecord(ctx->ev_copy_join, side);
            if (ew == hipSuccess) ew = hipStreamWaitEvent(s, ctx->ev_copy_join, 0);
            if (ew != hipSuccess) return side_fail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(ew)));
        } else {
        std::vector<CopySeg> segs = {{mb->start, m_start, nm * 8}, {mb->end, m_end, nm * 8}, {mb->in_off, db->d_m_in_off, nm * mk * 8}, {mb->in_cnt, db->d_m_in_cnt, nm * mk * 4},
                                     {mb->contig_idx, m_contig, has_contig ? nm * 4 : 0}, {mb->var_pos, d_pos, nv * 8}, {mb->a0_off, d_a0o, nv * 8}, {mb->a1_off, d_a1o, nv * 8},
                                     {mb->a0_len, d_a0l, nv * 4}, {mb->a1_len, d_a1l, nv * 4}, {mb->var_raw_space, d_raw, has_raw ? nv * 4 : 0}, {mb->var_type, d_type, nv},
                                     {mb->var_zyg, d_zyg, nv}, {mb->allele_bytes, d_alleles, alen}};
        rc = copy_in(ctx, segs);
        if (rc) return bail(rc);
        }
        dpk::DpPairs c;
        memset(&c, 0, sizeof(c));
        c.contig_idx = m_contig, c.start = m_start, c.end = m_end, c.in_off = db->d_m_in_off, c.in_cnt = db->d_m_in_cnt, c.n_multi = nm, c.k = mk, c.ppr = mppr;
        c.w_contig = d_contig, c.w_t_cnt = db->d_in_t_cnt, c.w_q_cnt = db->d_in_q_cnt, c.w_start = d_start, c.w_end = d_end, c.w_t_off = db->d_in_t_off, c.w_q_off = db->d_in_q_off;
        if (n) {
            hipLaunchKernelGGL(avk_dp_expand_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c);
            hipError_t ew = hipGetLastError();
            if (ew != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(ew)));
        }
    } else { /* the compact arrays as they are, then one kernel that writes the wide ones */
        dpk::DpCompact c;
        memset(&c, 0, sizeof(c));
        uint32_t *c_contig = has_contig ? (uint32_t *)tmp((n + 1) * 4) : nullptr, *c_start = (uint32_t *)tmp((n + 1) * 4), *c_len = (uint32_t *)tmp((n + 1) * 4),
                 *c_voff = (uint32_t *)tmp((n + 1) * 4), *c_pos = (uint32_t *)tmp((nv + 1) * 4), *c_aoff = (uint32_t *)tmp((nv + 1) * 4);
        uint16_t *c_tc = (uint16_t *)tmp((n + 1) * 2), *c_qc = (uint16_t *)tmp((n + 1) * 2);
        uint8_t *c_tz = (uint8_t *)tmp(nv + 16);
        if (rc) return bail(rc);
        /* a0_len / a1_len / raw_space have the wide arrays' type: they are copied straight into them */
        std::vector<CopySeg> segs = {{cb->start, c_start, n * 4}, {cb->len, c_len, n * 4}, {cb->v_off, c_voff, n * 4}, {cb->t_cnt, c_tc, n * 2}, {cb->q_cnt, c_qc, n * 2},
                                     {cb->contig_idx, c_contig, has_contig ? n * 4 : 0}, {cb->var_pos, c_pos, nv * 4}, {cb->a_off, c_aoff, nv * 4}, {cb->var_type_zyg, c_tz, nv},
                                     {cb->a0_len, d_a0l, nv * 4}, {cb->a1_len, d_a1l, nv * 4}, {cb->var_raw_space, d_raw, has_raw ? nv * 4 : 0}, {cb->allele_bytes, d_alleles, alen}};
        rc = copy_in(ctx, segs);
        if (rc) return bail(rc);
        c.contig_idx = c_contig, c.start = c_start, c.len = c_len, c.v_off = c_voff, c.t_cnt = c_tc, c.q_cnt = c_qc, c.var_pos = c_pos, c.a_off = c_aoff, c.a0_len = d_a0l, c.a1_len = d_a1l,
        c.var_raw = nullptr, c.var_type_zyg = c_tz, c.n_regions = n, c.n_variants = nv;
        c.w_contig = d_contig, c.w_t_cnt = db->d_in_t_cnt, c.w_q_cnt = db->d_in_q_cnt, c.w_a0_len = d_a0l, c.w_a1_len = d_a1l, c.w_raw = nullptr, c.w_start = d_start, c.w_end = d_end,
        c.w_t_off = db->d_in_t_off, c.w_q_off = db->d_in_q_off, c.w_pos = d_pos, c.w_a0_off = d_a0o, c.w_a1_off = d_a1o, c.w_type = d_type, c.w_zyg = d_zyg;
        const uint64_t m = n > nv ? n : nv;
        if (m) {
            hipLaunchKernelGGL(avk_dp_widen_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, c);
            hipError_t ew = hipGetLastError();
            if (ew != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(ew)));
        }
    }
    const auto t_copy = now();
    a.in.contig_idx = d_contig, a.in.start = d_start, a.in.end = d_end, a.in.t_off = db->d_in_t_off, a.in.q_off = db->d_in_q_off, a.in.t_cnt = db->d_in_t_cnt,
    a.in.q_cnt = db->d_in_q_cnt, a.in.var_pos = d_pos, a.in.var_type = d_type, a.in.var_zyg = d_zyg, a.in.var_raw = d_raw, a.in.a0_off = d_a0o, a.in.a1_off = d_a1o,
    a.in.a0_len = d_a0l, a.in.a1_len = d_a1l, a.in.alleles = d_alleles, a.in.n_regions = n, a.in.n_variants = nv, a.in.alleles_len = alen,
    a.in.contig_base = ctx->d_contig_tab, a.in.contig_len = ctx->d_contig_tab + ctx->contig_len.size(), a.in.n_contigs = (uint32_t)ctx->contig_len.size(),
    a.in.pairs_mode = pairs_mode ? 1u : 0u;
    { /* the calls this batch owns, guessed from its first and last region (batches of one job may share the call arrays: compare_main.cpp);
       * dp_region notes any region outside the guess */
        uint64_t lo = 0, hi = 0;
        if (pk) {
            hi = nv; /* the packed form owns its calls by construction */
        } else if (n && b) {
            const uint64_t a0 = b->t_cnt[0] ? b->t_off[0] : b->q_off[0], a1 = b->q_cnt[0] ? b->q_off[0] : b->t_off[0];
            lo = a0 < a1 ? a0 : a1;
            const uint64_t e0 = b->t_off[n - 1] + b->t_cnt[n - 1], e1 = b->q_off[n - 1] + b->q_cnt[n - 1];
            hi = e0 > e1 ? e0 : e1;
        } else if (n && cb) {
            lo = cb->v_off[0];
            hi = (uint64_t)cb->v_off[n - 1] + cb->t_cnt[n - 1] + cb->q_cnt[n - 1];
        } else
            hi = nv; /* (the pair form has no per-call outputs) */
        if (hi > nv) hi = nv;
        if (lo > hi) lo = hi;
        a.in.v_lo = lo, a.in.v_hi = hi;
        if ((b || cb) && !pairs_mode && hi > lo) { /* explicit offsets: ownership is counted (DpIn::owned) */
            a.in.owned = (uint8_t *)tmp(hi - lo + 16);
            if (rc) return bail(rc);
            if (hipMemsetAsync(a.in.owned, 0, hi - lo, s) != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(hipGetLastError())));
        }
    }
    const bool lanes = ctx->lane_kernel && ctx->use_packed_reference && ctx->d_ref2b;
    a.opt.tier0_bytes = avk::bulk_slice_bytes((uint64_t)ctx->lds_bytes_per_wave), a.opt.tier0_ed_cap = (uint32_t)ctx->lds_ed_cap, a.opt.tier1_bytes = (uint64_t)ctx->lds2_bytes_per_wave,
    a.opt.tier1_ed_cap = (uint32_t)ctx->lds2_ed_cap, a.opt.solo_min_variants = pairs_mode && !ctx->pair_classes ? 0u : (uint32_t)ctx->solo_min_variants, a.opt.max_branch = 50,
    a.opt.class_c_nodes_x2 = (uint32_t)ctx->class_c_nodes_x2, a.opt.lane_min_regions = lanes ? (uint64_t)ctx->lane_min_regions : 0xFFFFFFFFull,
    a.opt.lane_max_calls = (uint32_t)ctx->lane_max_calls, a.opt.lane_min_batch = (uint64_t)ctx->lane_min_batch, a.opt.lane_max_est = (uint32_t)ctx->lane_max_est;
    a.opt.stripe_w = ctx->lane_stripe ? (uint32_t)ctx->lane_head_width : 0u;
    a.opt.lane_pairs = ctx->lane_pairs ? 1u : 0u;
    a.opt.head_est = (uint32_t)ctx->lane_head_est, a.opt.het_min = (uint32_t)ctx->het_search_min;
    a.opt.class_c_below = (uint64_t)ctx->class_c_below;
    hipError_t e = variant_done ? hipSuccess : hipMemsetAsync(a.st, 0, sizeof(dpk::DpState), s);
    if (e == hipSuccess && nv && !variant_done) {
        hipLaunchKernelGGL(avk_dp_variant_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, a);
        e = hipGetLastError();
    }
    dpk::DpState *hs = (dpk::DpState *)ctx->h_dpstate;
    auto region_passes = [&]() -> hipError_t { /* everything that depends on alt_ed, up to the state block on the host */
        hipError_t x = hipSuccess;
        if (n) {
            hipLaunchKernelGGL(avk_dp_region_kernel, dim3(n_blocks), dim3(256), 0, s, a, d_block_sums);
            if (a.in.owned && a.in.v_hi > a.in.v_lo)
                hipLaunchKernelGGL(avk_dp_count_owned_kernel, dim3((unsigned)((a.in.v_hi - a.in.v_lo + 4095) / 4096)), dim3(256), 0, s, (const uint8_t *)a.in.owned, a.in.v_hi - a.in.v_lo, a.st);
            hipLaunchKernelGGL(avk_dp_scan_blocks_kernel, dim3(AVK_DP_BS), dim3(1024), 0, s, d_block_sums, n_blocks, a.st);
            hipLaunchKernelGGL(avk_dp_scan_apply_kernel, dim3(n_blocks), dim3(256), 0, s, a, (const uint64_t *)d_block_sums);
            hipLaunchKernelGGL(avk_dp_lane_switch_kernel, dim3(1), dim3(64), 0, s, a);
            hipLaunchKernelGGL(avk_dp_hist_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, a);
            hipLaunchKernelGGL(avk_dp_bucket_bases_kernel, dim3(1), dim3(256), 0, s, a);
            hipLaunchKernelGGL(avk_dp_scatter_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, a);
            x = hipGetLastError();
        }
        mark(2);
        if (x == hipSuccess) x = hipMemcpyAsync(hs, a.st, sizeof(dpk::DpState), hipMemcpyDeviceToHost, s);
        if (x == hipSuccess && pk_totals) x = hipMemcpyAsync(hs + 1, pk_totals, 16, hipMemcpyDeviceToHost, s); /* (the packed forms' two sums ride along: 16 bytes behind the state block) */
        if (x == hipSuccess) x = hipStreamSynchronize(s);
        if (x == hipSuccess) engine_rate_check(ctx);
        return x;
    };
    if (e == hipSuccess) e = region_passes();
    if (e == hipSuccess && (pk || pm)) { /* the two sums the packed form implies must be what the caller says they are */
        uint64_t tot[2] = {0, 0};
        memcpy(tot, hs + 1, sizeof(tot));
        if ((((pm ? pm->n_regions : n) && tot[0] != nv) || (nv && tot[1] != alen)))
            return bail(fail(ctx, AVK_E_ARG, "packed batch: the call counts sum to %llu (n_variants %llu), the allele lengths to %llu (allele_bytes_len %llu)",
                             (unsigned long long)tot[0], (unsigned long long)nv, (unsigned long long)tot[1], (unsigned long long)alen));
    }
    std::vector<uint64_t> pk_aoff; /* packed form: allele offsets on the host, made only when a call needs the host's edit distance */
    const uint8_t *pk_l0 = pk ? pk->a0_len : (pm ? pm->a0_len : nullptr), *pk_l1 = pk ? pk->a1_len : (pm ? pm->a1_len : nullptr);
    if (e == hipSuccess && hs->n_pending && pk_l0) {
        pk_aoff.resize(nv + 1);
        uint64_t run = 0;
        for (uint64_t v = 0; v < nv; ++v) pk_aoff[v] = run, run += (uint64_t)pk_l0[v] + pk_l1[v];
    }
    if (e == hipSuccess && hs->n_pending) {
        /* calls whose two alleles are both long after the common prefix and suffix are gone: their alt_ed comes from the host
         * (avk_edit_distance, the routine the host-side packer uses for every call), and the region passes run again */
        const uint32_t np = hs->n_pending;
        std::vector<uint32_t> idx(np), ed(np);
        e = hipMemcpy(idx.data(), a.pending, (size_t)np * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            avk_parallel_for(np, avk_host_threads(), [&](unsigned, uint64_t lo, uint64_t hi) {
                for (uint64_t k = lo; k < hi; ++k) {
                    const uint64_t v = idx[k];
                    const uint64_t o0 = pk_l0 ? pk_aoff[v] : (b ? b->a0_off[v] : (cb ? cb->a_off[v] : mb->a0_off[v])), l0 = pk_l0 ? pk_l0[v] : (b ? b->a0_len[v] : (cb ? cb->a0_len[v] : mb->a0_len[v])),
                                   o1 = pk_l0 ? o0 + l0 : (b ? b->a1_off[v] : (cb ? o0 + l0 : mb->a1_off[v])), l1 = pk_l0 ? pk_l1[v] : (b ? b->a1_len[v] : (cb ? cb->a1_len[v] : mb->a1_len[v]));
                    ed[k] = (uint32_t)avk::host_edit_distance(host_alleles + o0, l0, host_alleles + o1, l1);
                }
            });
            uint32_t *d_idx = (uint32_t *)tmp((size_t)np * 4), *d_ed = (uint32_t *)tmp((size_t)np * 4);
            if (rc) return bail(rc);
            e = hipMemcpy(d_idx, idx.data(), (size_t)np * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_ed, ed.data(), (size_t)np * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(avk_dp_patch_ed_kernel, dim3((np + 255) / 256), dim3(256), 0, s, a.vinfo, (const uint32_t *)d_idx, (const uint32_t *)d_ed, np);
                e = hipMemsetAsync(a.st, 0, sizeof(dpk::DpState), s);
            }
            if (e == hipSuccess) e = region_passes();
        }
    }
    if (e != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(e)));
    const auto t_plan = now();
    if (hs->err & dpk::DP_ERR_RANGE) return bail(fail(ctx, AVK_E_ARG, "variant range of a region exceeds n_variants"));
    if (hs->err & dpk::DP_ERR_ALLELE) return bail(fail(ctx, AVK_E_ARG, "allele range exceeds allele_bytes_len"));
    if (hs->err & dpk::DP_ERR_BLOB) return bail(fail(ctx, AVK_E_ARG, "region blob exceeds 2 GiB; split the region's alleles"));
    if (hs->total_v > 0x7FFFFFFFull) return bail(fail(ctx, AVK_E_ARG, "more than 2^31 variant records; split the batch"));
    db->v_lo = a.in.v_lo, db->v_hi = a.in.v_hi;
    if (hs->err & dpk::DP_NOTE_OUTSIDE) db->v_lo = 0, db->v_hi = nv;
    /* every call of the range is owned, and once: nothing to preserve.  The packed forms' offsets are the running sums of the counts — dense when the sums are right;
     * forms with explicit offsets are dense when as many calls are marked as the range has and the counts add up to the same number */
    db->var_dense = !(hs->err & dpk::DP_NOTE_OUTSIDE) && hs->total_v == db->v_hi - db->v_lo && (!a.in.owned || hs->n_owned == db->v_hi - db->v_lo);
    if (hs->total_blob_words / 2 > 0xFFFFFFFFull) return bail(fail(ctx, AVK_E_ARG, "region blob arena exceeds its limits; split the batch"));
    db->n_variants_dev = hs->total_v;
    db->seq_total = hs->total_seq;
    db->n_bp_groups = hs->total_groups;
    { /* per-wave HBM slices of this batch's launches: large enough for 98 % of the regions predicted to need the tier (64 MB at most) */
        uint64_t n_c = 0, run = 0;
        for (int k = 0; k < dpk::DP_NEED_BUCKETS; ++k) n_c += hs->need_hist[k];
        int pick = 0;
        for (int k = 0; k < dpk::DP_NEED_BUCKETS && n_c >= 64; ++k) {
            run += hs->need_hist[k];
            pick = k;
            if (run * 100 >= n_c * 98) break;
        }
        if (pick > 6) pick = 6;
        const int64_t want = 1ll << (20 + pick);
        db->ws_bytes_eff = want > ctx->ws_bytes_per_wave && ctx->ws_bytes_per_wave > 0 && ctx->adaptive_ws ? want : 0;
        /* regions predicted beyond the last bucket (windows of tens of kilobases with dozens of calls): the shared slices start at the size of the library's first
         * capacity retry, so that these regions are solved in the step and not again, one sub-batch at a time, by avk_results_download */
        db->big_bytes_eff = hs->need_hist[dpk::DP_NEED_BUCKETS - 1] > 0 && ctx->adaptive_ws ? (1ll << 30) : 0;
    }
    db->plan.n_hbm = hs->n_hbm, db->plan.n_hard = hs->n_hard, db->plan.n_fast_total = hs->n_fast_total, db->plan.n_hbm_notwide = hs->n_hbm_notwide;
    uint32_t tiles_total = 0;
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) {
        db->plan.n_fast[fc] = hs->n_fast[fc], db->plan.n_fast_heavy[fc] = hs->n_fast_heavy[fc], db->plan.fast_base[fc] = hs->fast_base[fc];
        db->fast_word_base[fc] = hs->fast_word_base[fc], db->fast_tiles[fc] = hs->fast_tiles[fc];
        tiles_total += hs->fast_tiles[fc];
    }
    /* the batch's own buffers */
    const uint64_t nvd = hs->total_v;
    db->d_regions = (AvkDevRegion *)kept((n + 1) * sizeof(AvkDevRegion));
    db->d_blob = (uint32_t *)kept(((size_t)hs->total_blob_words + 4) * 4);
    db->d_region_out = (uint32_t *)kept((n * 4 + 4) * 4);
    db->d_var_out = (uint32_t *)kept((nvd + 1) * 4);
    db->d_seqlen = (uint32_t *)kept((n * 5 + 1) * 4);
    db->d_tally = (uint64_t *)kept((size_t)AVK_TALLY_STRIDE * 8);
    db->d_partials = (uint64_t *)kept((size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES * 8);
    db->d_counters = (uint32_t *)kept((size_t)AVK_N_COUNTERS * 4);
    db->d_overflow = (uint32_t *)kept((n + 1024) * 4);
    db->d_overflow2 = (uint32_t *)kept((n + 1) * 4);
    db->d_overflow3 = (uint32_t *)kept((3 * (n + 1) + 1024) * 4);
    db->d_overflow4 = (uint32_t *)kept((n + 1) * 4);
    db->d_overflow5 = (uint32_t *)kept((n + 1) * 4);
    db->d_overflow6 = (uint32_t *)kept((n + 1) * 4);
    db->d_overflow7 = (uint32_t *)kept((n + 1) * 4);
    db->d_overflow8 = (uint32_t *)kept((n + 1) * 4);
    db->d_notwide = (uint32_t *)kept((n + 2) * 4);
    if (hs->n_fast_total) db->d_fast = (uint32_t *)kept(((size_t)hs->fast_words + 64) * 4);
    /* the packer's arguments in device memory: the waves that solve handed-back regions write their records themselves */
    if (hs->n_fast_total) db->d_dp_args = (dpk::DpArgs *)kept(sizeof(dpk::DpArgs));
    if (rc) return bail(rc);
    a.regions = db->d_regions, a.blob = db->d_blob, a.fast = db->d_fast;
    db->dp_args = a;
    const bool clear_beside = n >= 262144; /* (a small batch: the two events between the streams cost more than the fills, 1.16 instead of 0.98 ms per 46,000-region call) */
    if (clear_beside) { /* the batch's partial tallies and counters are cleared beside the writers, on a side stream (in front of the solver launches the two fills took 55 us) */
        hipStream_t side = (ctx->up_side ? ctx->up_side : ctx->lane_stream4);
        hipError_t ez = hipEventRecord(ctx->ev_copy_fork, s); /* the buffers may have been another batch's until here */
        if (ez == hipSuccess) ez = hipStreamWaitEvent(side, ctx->ev_copy_fork, 0);
        if (ez == hipSuccess && db->d_dp_args) ez = hipMemcpyAsync(db->d_dp_args, &db->dp_args, sizeof(dpk::DpArgs), hipMemcpyHostToDevice, side); /* (behind the writers it was
                                                                                                     most of a 130 us gap in front of the solver launches) */
        if (ez == hipSuccess) ez = hipMemsetAsync(db->d_partials, 0, (size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES * sizeof(uint64_t), side);
        if (ez == hipSuccess) ez = hipMemsetAsync(db->d_counters, 0, AVK_N_COUNTERS * sizeof(uint32_t), side);
        if (ez == hipSuccess) ez = hipEventRecord(ctx->ev_copy_join, side);
        if (ez != hipSuccess) {
            (void)hipStreamSynchronize(side);
            return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(ez)));
        }
        db->scratch_clean = true;
    }
    if (tiles_total) hipLaunchKernelGGL(avk_dp_fast_records_kernel, dim3((tiles_total + 3) / 4), dim3(256), 0, s, a, tiles_total);
    /* records and blobs: now for the regions the wave-per-region launches start with; for the lanes' regions when (and if) a launch asks for them */
    const uint32_t n_eager = (uint32_t)n - hs->n_fast_total;
    if (n_eager) {
        hipLaunchKernelGGL(avk_dp_region_records_kernel, dim3((n_eager + 255) / 256), dim3(256), 0, s, a, n_eager);
        hipLaunchKernelGGL(avk_dp_region_records_wave_kernel, dim3(256), dim3(256), 0, s, a);
    }
    db->records_full = hs->n_fast_total == 0;
    db->lazy_from = n_eager;
    if (db->d_dp_args && !clear_beside) {
        hipError_t ec = hipMemcpyAsync(db->d_dp_args, &db->dp_args, sizeof(dpk::DpArgs), hipMemcpyHostToDevice, s);
        if (ec != hipSuccess) return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(ec)));
    }
    e = hipGetLastError();
    if (e == hipSuccess && clear_beside) e = hipStreamWaitEvent(s, ctx->ev_copy_join, 0); /* the cleared scratch, before anything that follows on this stream */
    if (e != hipSuccess) {
        (void)hipStreamSynchronize((ctx->up_side ? ctx->up_side : ctx->lane_stream4));
        return bail(fail(ctx, AVK_E_HIP, "device packing failed: %s", hipGetErrorString(e)));
    }
    mark(3);
    for (void *p : temps) pool_release(ctx, p); /* in stream order: the writers above run before anything that is handed these buffers next */
    if (!ctx->up_stream && ctx->ev_pool_fence) { /* ... which an upload on the packing stream of the asynchronous boundary is not, by itself: it waits for this point once */
        if (hipEventRecord(ctx->ev_pool_fence, s) == hipSuccess) ctx->pool_fence_pending = true;
        else (void)hipGetLastError();
    }
    if (timing)
        fprintf(stderr, "avk upload (device-packed): %llu regions, %llu calls: buffers %.3f ms, copies queued %.3f ms, packing kernels + plan %.3f ms, writers queued %.3f ms; lanes %u regions in %u tiles, class C %u, class B %u\n",
                (unsigned long long)n, (unsigned long long)nv, ms(t_start, t_alloc), ms(t_alloc, t_copy), ms(t_copy, t_plan), ms(t_plan, now()), hs->n_fast_total, tiles_total,
                hs->n_hbm, hs->n_hard);
    *out = db;
    return 0;
}

/* ---- download of a device-packed batch: dp_unpack on the device, then plain copies into the caller's arrays ------------------------- */
/* `later` (avk_compare_packed_submit): the copies out go on the context's copy-out stream behind `later->ev_unpacked`, the tally into `later->h_tally`, nothing is
 * waited for and the device buffers of the caller's layout stay in `later->temps` until avk_wait has seen `later->ev_done` (every destination must be pinned) */
struct DownloadLater {
    std::vector<void *> temps;
    uint64_t *h_tally;
    hipEvent_t ev_unpacked, ev_done;
    /* the parts of ONE split call (compare_packed_split) spill their BASEPAIR groups into one device buffer behind one counter: a part's dp_unpack appends where the
     * part before it stopped, the words it writes index the call's one list, and the caller reads the count and copies the list once, after the last part */
    uint32_t *shared_spill = nullptr, *shared_spill_count = nullptr;
};
static int download_device_packed(avk_ctx *ctx, avk_dev_batch *db, avk_result_batch *out, uint8_t *pair_exact, uint64_t *tally_words /* [AVK_TALLY_STRIDE] */,
                                  DownloadLater *later = nullptr) {
    const uint64_t n = db->n_regions, nv = db->n_variants_host;
    hipStream_t s = ctx->stream;
    std::vector<void *> temps;
    int rc = 0;
    auto tmp = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (!rc) rc = pool_alloc(ctx, &p, bytes);
        if (p) temps.push_back(p);
        return p;
    };
    auto done = [&](int code) {
        for (void *p : temps) pool_release(ctx, p);
        return code;
    };
    dpk::DpOut o;
    memset(&o, 0, sizeof(o));
    o.region_out = db->d_region_out, o.var_out = db->d_var_out, o.v_off = db->d_voff, o.t_off = db->d_in_t_off, o.q_off = db->d_in_q_off, o.t_cnt = db->d_in_t_cnt,
    o.q_cnt = db->d_in_q_cnt, o.n_regions = n, o.n_variants = nv, o.mode = db->last_mode;
    o.pk_voff = db->d_pk_voff, o.pk_tc = db->d_pk_tc, o.pk_qc = db->d_pk_qc;
    if (out->status || !out->region_packed) o.status = (int32_t *)tmp((n + 1) * 4);
    if (out->region_packed) o.region_packed = (uint64_t *)tmp((n + 1) * 8);
    if (out->ed_h1) o.ed_h1 = (uint32_t *)tmp((n + 1) * 4);
    if (out->ed_h2) o.ed_h2 = (uint32_t *)tmp((n + 1) * 4);
    if (out->n_optima) o.n_optima = (uint32_t *)tmp((n + 1) * 4);
    if (out->type_present) o.type_present = (uint16_t *)tmp((n + 1) * 2);
    const bool want_var = db->last_mode == 0 && (out->var_expected || out->var_observed || out->var_class || out->var_zyg || out->var_packed) && db->v_hi > db->v_lo;
    const uint64_t nvr = db->v_hi - db->v_lo; /* the calls this batch owns: only they are copied back */
    o.v_lo = db->v_lo;
    if (want_var) {
        if (out->var_expected) o.var_expected = (uint8_t *)tmp(nvr + 16);
        if (out->var_observed) o.var_observed = (uint8_t *)tmp(nvr + 16);
        if (out->var_class) o.var_class = (uint8_t *)tmp(nvr + 16);
        if (out->var_zyg) o.var_zyg = (uint8_t *)tmp(nvr + 16);
        if (out->var_packed) o.var_packed = (uint8_t *)tmp(nvr + 16);
    }
    uint8_t *d_exact = pair_exact ? (uint8_t *)tmp(n + 16) : nullptr;
    const bool want_bp_packed = out->bp_packed && out->bp_spilled && out->bp_groups && db->d_bp && db->d_bp_off && db->last_mode == 0;
    const bool bp_shared = later && later->shared_spill && later->shared_spill_count;
    if (want_bp_packed) {
        if (later && !bp_shared) return done(fail(ctx, AVK_E_STATE, "the packed BASEPAIR groups need a synchronous call"));
        o.bp_off_dev = db->d_bp_off, o.bp_dev = db->d_bp;
        o.bp_packed = (uint32_t *)tmp((n + 1) * 4);
        if (bp_shared) o.bp_spill = later->shared_spill, o.bp_spill_count = later->shared_spill_count;
        else {
            o.bp_spill = (uint32_t *)tmp(((size_t)db->n_bp_groups + 1) * 16);
            o.bp_spill_count = (uint32_t *)tmp(256);
            if (!rc && hipMemsetAsync(o.bp_spill_count, 0, 4, s) != hipSuccess) rc = fail(ctx, AVK_E_HIP, "result unpacking failed: %s", hipGetErrorString(hipGetLastError()));
        }
    }
    if (rc) return done(rc);
    if (!ctx->h_dpstate) {
        hipError_t e = hipHostMalloc((void **)&ctx->h_dpstate, sizeof(dpk::DpState) + 16, hipHostMallocDefault);
        if (e != hipSuccess) return done(fail(ctx, AVK_E_HIP, "pinned state block: %s", hipGetErrorString(e)));
    }
    static_assert(sizeof(dpk::DpState) >= AVK_TALLY_STRIDE * 8, "the pinned state block also receives the tally");
    hipError_t e = hipSuccess;
    if (want_var && !db->var_dense) { /* calls of the range that no region of this batch owns keep what the caller's arrays hold (another batch of the job may own them) */
        std::vector<CopySeg> pre = {{out->var_expected ? out->var_expected + db->v_lo : nullptr, o.var_expected, out->var_expected ? nvr : 0},
                                    {out->var_observed ? out->var_observed + db->v_lo : nullptr, o.var_observed, out->var_observed ? nvr : 0},
                                    {out->var_class ? out->var_class + db->v_lo : nullptr, o.var_class, out->var_class ? nvr : 0},
                                    {out->var_zyg ? out->var_zyg + db->v_lo : nullptr, o.var_zyg, out->var_zyg ? nvr : 0},
                                    {out->var_packed ? out->var_packed + db->v_lo : nullptr, o.var_packed, out->var_packed ? nvr : 0}};
        rc = copy_in(ctx, pre);
        if (rc) return done(rc);
    }
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(avk_dp_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o, d_exact);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return done(fail(ctx, AVK_E_HIP, "result unpacking failed: %s", hipGetErrorString(e)));
    std::vector<CopySeg> segs = {{out->status, o.status, out->status ? n * 4 : 0}, {out->region_packed, o.region_packed, out->region_packed ? n * 8 : 0},
                                 {out->ed_h1, o.ed_h1, n * 4}, {out->ed_h2, o.ed_h2, n * 4}, {out->n_optima, o.n_optima, n * 4},
                                 {out->type_present, o.type_present, n * 2}, {pair_exact, d_exact, pair_exact ? n : 0}};
    if (want_var) {
        segs.push_back({out->var_expected ? out->var_expected + db->v_lo : nullptr, o.var_expected, out->var_expected ? nvr : 0});
        segs.push_back({out->var_observed ? out->var_observed + db->v_lo : nullptr, o.var_observed, out->var_observed ? nvr : 0});
        segs.push_back({out->var_class ? out->var_class + db->v_lo : nullptr, o.var_class, out->var_class ? nvr : 0});
        segs.push_back({out->var_zyg ? out->var_zyg + db->v_lo : nullptr, o.var_zyg, out->var_zyg ? nvr : 0});
        segs.push_back({out->var_packed ? out->var_packed + db->v_lo : nullptr, o.var_packed, out->var_packed ? nvr : 0});
    }
    if (out->group_metrics && ctx->emit_group_metrics && db->d_gm) segs.push_back({out->group_metrics, db->d_gm, n * AVK_N_GROUPS * AVK_N_FIELDS * sizeof(uint32_t)});
    if (want_bp_packed) segs.push_back({out->bp_packed, o.bp_packed, n * sizeof(uint32_t)}); /* (the spilled groups and their count: below, behind the other copies — or the split call's, once) */
    else if (out->bp_off && out->bp_groups && db->d_bp && db->d_bp_off && db->last_mode == 0) {
        segs.push_back({out->bp_off, db->d_bp_off, (n + 1) * sizeof(uint32_t)});
        segs.push_back({out->bp_groups, db->d_bp, (size_t)db->n_bp_groups * 4 * sizeof(uint32_t)});
    }
    if (later) { /* queued on the copy-out stream, finished by avk_wait */
        hipStream_t so = ctx->copy_out_stream;
        hipError_t el = hipEventRecord(later->ev_unpacked, s);
        if (el == hipSuccess) el = hipStreamWaitEvent(so, later->ev_unpacked, 0);
        for (const CopySeg &sg : segs)
            if (el == hipSuccess && sg.bytes && sg.host) el = hipMemcpyAsync((void *)sg.host, sg.dev, sg.bytes, hipMemcpyDeviceToHost, so);
        if (el == hipSuccess) el = hipMemcpyAsync(later->h_tally, db->d_tally, (size_t)AVK_TALLY_STRIDE * 8, hipMemcpyDeviceToHost, so);
        if (el == hipSuccess) el = hipEventRecord(later->ev_done, so);
        if (el != hipSuccess) {
            (void)hipStreamSynchronize(so);
            return done(fail(ctx, AVK_E_HIP, "queueing the results' copies failed: %s", hipGetErrorString(el)));
        }
        later->temps.swap(temps);
        return 0;
    }
    CopyOut co;
    rc = copy_out(ctx, segs, &co);
    if (rc) return done(rc);
    e = hipMemcpyAsync(ctx->h_dpstate, db->d_tally, (size_t)AVK_TALLY_STRIDE * 8, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return done(fail(ctx, AVK_E_HIP, "tally download failed: %s", hipGetErrorString(e)));
    if (want_bp_packed) {
        /* how many groups were spilled decides how much of the list is copied.  The count rides behind the copies above (they start right behind dp_unpack, without
         * a word from the host), the list — a tenth of the results — behind the count: round 5 read the count first and every copy waited for that round trip */
        uint32_t *h_count = (uint32_t *)((uint8_t *)ctx->h_dpstate + sizeof(dpk::DpState));
        e = hipMemcpyAsync(h_count, o.bp_spill_count, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return done(fail(ctx, AVK_E_HIP, "result unpacking failed: %s", hipGetErrorString(e)));
        const uint32_t spilled = *h_count;
        out->bp_spilled[0] = spilled;
        if (spilled) { /* (pinned: queued, finish_copy_out waits for it; pageable: a blocking copy — the bounce buffer holds the parts queued above) */
            const size_t tail_bytes = (size_t)spilled * 4 * sizeof(uint32_t);
            e = is_pinned(out->bp_groups, tail_bytes) ? (copies_by_kernel(ctx) ? kernel_copy(ctx, out->bp_groups, o.bp_spill, tail_bytes, s)
                                                                            : hipMemcpyAsync(out->bp_groups, o.bp_spill, tail_bytes, hipMemcpyDeviceToHost, s))
                                                      : hipMemcpy(out->bp_groups, o.bp_spill, tail_bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return done(fail(ctx, AVK_E_HIP, "result unpacking failed: %s", hipGetErrorString(e)));
        }
    }
    if (getenv("AVK_TIMING")) { /* the last mark of the call's device timeline: behind the copies out */
        if (!ctx->ev_tl[4] && hipEventCreate(&ctx->ev_tl[4]) != hipSuccess) ctx->ev_tl[4] = nullptr, (void)hipGetLastError();
        if (ctx->ev_tl[4]) (void)hipEventRecord(ctx->ev_tl[4], s);
    }
    rc = finish_copy_out(ctx, co);
    if (rc) return done(rc);
    memcpy(tally_words, ctx->h_dpstate, (size_t)AVK_TALLY_STRIDE * 8);
    return done(0);
}

/* every region record and blob of a device-packed batch (a run without lane launches, the host-side view) */
static int ensure_all_records(avk_ctx *ctx, avk_dev_batch *db, hipStream_t s) {
    if (!db->dev_packed || db->records_full || !db->n_regions) return 0;
    const uint32_t n = (uint32_t)db->n_regions;
    /* n_big restarts: the wave-level writer takes the large regions of this pass */
    AVK_HIP(ctx, hipMemsetAsync(&db->dp_args.st->n_big, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(avk_dp_region_records_kernel, dim3((n + 255) / 256), dim3(256), 0, s, db->dp_args, n);
    hipLaunchKernelGGL(avk_dp_region_records_wave_kernel, dim3(256), dim3(256), 0, s, db->dp_args);
    AVK_HIP(ctx, hipGetLastError());
    db->records_full = true;
    return 0;
}

/* The host-side view of a device-packed batch (region records, blobs, the map from per-call output words to the caller's calls), fetched only when
 * something needs it: the capacity retry and the sequence outputs of avk_results_download. */
static int materialize_host_view(avk_ctx *ctx, avk_dev_batch *db) {
    if (!db->dev_packed || !db->host.regions.empty() || !db->n_regions) return 0;
    const uint64_t n = db->n_regions;
    {
        const int rf = ensure_all_records(ctx, db, ctx->stream);
        if (rf) return rf;
        AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    avk::PodVec<AvkDevRegion> recs;
    recs.resize(n);
    AVK_HIP(ctx, hipMemcpy(recs.data(), db->d_regions, n * sizeof(AvkDevRegion), hipMemcpyDeviceToHost));
    db->host.regions.resize(n);
    uint64_t blob_words = 2;
    for (uint64_t k = 0; k < n; ++k) {
        db->host.regions[recs[k].orig] = recs[k];
        const uint64_t end = 2ull * recs[k].blob_off + recs[k].blob_bytes / 4;
        if (recs[k].blob_bytes && end > blob_words) blob_words = end;
    }
    db->host.blob.resize(blob_words);
    AVK_HIP(ctx, hipMemcpy(db->host.blob.data(), db->d_blob, blob_words * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> toff(n), qoff(n);
    if (db->d_pk_voff) { /* read from its packed source: the running sum of the calls, the query calls behind the truth calls */
        std::vector<uint8_t> tcs(n);
        AVK_HIP(ctx, hipMemcpy(toff.data(), db->d_pk_voff, n * 8, hipMemcpyDeviceToHost));
        AVK_HIP(ctx, hipMemcpy(tcs.data(), db->d_pk_tc, n, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < n; ++r) qoff[r] = toff[r] + tcs[r];
    } else {
        AVK_HIP(ctx, hipMemcpy(toff.data(), db->d_in_t_off, n * 8, hipMemcpyDeviceToHost));
        AVK_HIP(ctx, hipMemcpy(qoff.data(), db->d_in_q_off, n * 8, hipMemcpyDeviceToHost));
    }
    db->host.dev2host.resize(db->n_variants_dev);
    for (uint64_t r = 0; r < n; ++r) {
        const AvkDevRegion &dr = db->host.regions[r];
        for (uint32_t k = 0; k < dr.t_cnt + dr.q_cnt; ++k) db->host.dev2host[dr.v_off + k] = k < dr.t_cnt ? toff[r] + k : qoff[r] + (k - dr.t_cnt);
    }
    return 0;
}
