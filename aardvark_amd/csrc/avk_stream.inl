/*
 * avk_stream.inl — the one-shot form of the boundary (avk_compare_batch) for large batches: host arrays in, host arrays out, the
 * region batch never exists twice on the host.  Included by avk_host.hip (it uses the context, the kernels and upload / run / download
 * of the resident path).
 *
 * The reference's loop (src/main.rs:251-268) hands one CompareRegion at a time to solve_compare_region.  Here the caller's
 * structure-of-arrays batch is cut three ways:
 *   fast      regions of the lane-per-region classes (avk_dev_types.h, about 98 % of a genome): classified and written STRAIGHT into
 *             pinned fast records by the host threads, tile range by tile range, each range copied to the device while the next one is
 *             written; one set of lane-kernel launches at the end;
 *   general   everything else: gathered into a small batch and sent through the resident path (pack_batch -> wave-per-region kernels)
 *             FIRST, so its kernels run while the fast records are still being written;
 *   deferred  fast regions a lane handed back (an edit distance beyond the lane cap on the best path, a non-ACGT window): their indices
 *             come back with the results; they are gathered and solved by the wave-per-region kernels in a second small call.
 * Results: 16 bytes per region and one word per variant come back into pinned memory and are unpacked into the caller's arrays by the
 * host threads.  Outputs are identical to the resident path's (tests/test_gpu_lane.py::test_one_shot_path_equals_resident_path).
 */

namespace {

/* A small persistent pool: the one-shot path runs five or six parallel loops per call, and starting 16 threads for each costs more than a
 * chr20-sized call's kernels.  Workers sleep on a condition variable between loops; one loop at a time (calls on different contexts
 * take turns). */
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
            /* the loops of one call follow each other within microseconds: look for the next one for a while before sleeping */
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
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if (const char *e = getenv("AVK_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 256) nt = (unsigned)v;
    }
    return nt;
}

/* grow-only staging buffers of the one-shot path (kept by the context: pinning and device allocation cost more than a call) */
struct StreamBufs {
    uint32_t *h_fast = nullptr, *d_fast = nullptr;
    size_t fast_words = 0;
    uint32_t *h_rout = nullptr, *d_rout = nullptr;
    size_t rout_words = 0;
    uint32_t *h_vout = nullptr, *d_vout = nullptr;
    size_t vout_words = 0;
    uint32_t *d_defer = nullptr, *h_defer = nullptr;
    size_t defer_words = 0;
    uint64_t *d_partials = nullptr, *d_tally = nullptr, *h_tally = nullptr;
    uint32_t *d_counters = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied = nullptr, ev_lane_done = nullptr, ev_lane_done2 = nullptr, ev_lane_done3 = nullptr;
};

void stream_bufs_free(StreamBufs *b) {
    if (!b) return;
    if (b->h_fast) (void)hipHostFree(b->h_fast);
    if (b->h_rout) (void)hipHostFree(b->h_rout);
    if (b->h_vout) (void)hipHostFree(b->h_vout);
    if (b->h_defer) (void)hipHostFree(b->h_defer);
    if (b->h_tally) (void)hipHostFree(b->h_tally);
    void *dev[] = {b->d_fast, b->d_rout, b->d_vout, b->d_defer, b->d_partials, b->d_tally, b->d_counters};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (b->ev_copied) (void)hipEventDestroy(b->ev_copied);
    if (b->ev_lane_done) (void)hipEventDestroy(b->ev_lane_done);
    if (b->ev_lane_done3) (void)hipEventDestroy(b->ev_lane_done3);
    if (b->ev_lane_done2) (void)hipEventDestroy(b->ev_lane_done2);
    if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
    delete b;
}

template <class T> int grow_pair(avk_ctx *ctx, T **h, T **d, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    need += need / 8 + 1024;
    if (*h) (void)hipHostFree(*h);
    if (*d) (void)hipFree(*d);
    *h = nullptr;
    *d = nullptr;
    *cap = 0;
    AVK_HIP(ctx, hipHostMalloc((void **)h, need * sizeof(T), hipHostMallocDefault));
    AVK_HIP(ctx, hipMalloc((void **)d, need * sizeof(T)));
    *cap = need;
    return 0;
}

/* what the classification pass finds out about a region */
struct StreamClass {
    std::vector<uint8_t> cls;     /* 0 = general path, 1 + k = fast class k */
    std::vector<uint8_t> key;     /* cost key inside the class (fast_cost_key): tiles hold regions of one cost */
    std::vector<uint8_t> alt_ed;  /* per caller variant: Variant::alt_ed (fast regions only) */
    std::vector<uint32_t> v_off;  /* first per-variant output word of a region (prefix sum of t_cnt + q_cnt) */
};

} // namespace

static int stream_bufs_init(avk_ctx *ctx) {
    if (ctx->sbufs) return 0;
    ctx->sbufs = new StreamBufs();
    AVK_HIP(ctx, hipStreamCreateWithFlags(&ctx->sbufs->copy_stream, hipStreamNonBlocking));
    AVK_HIP(ctx, hipEventCreateWithFlags(&ctx->sbufs->ev_copied, hipEventDisableTiming));
    AVK_HIP(ctx, hipEventCreateWithFlags(&ctx->sbufs->ev_lane_done, hipEventDisableTiming));
    AVK_HIP(ctx, hipEventCreateWithFlags(&ctx->sbufs->ev_lane_done3, hipEventDisableTiming));
    AVK_HIP(ctx, hipEventCreateWithFlags(&ctx->sbufs->ev_lane_done2, hipEventDisableTiming));
    AVK_HIP(ctx, hipMalloc((void **)&ctx->sbufs->d_partials, (size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES * sizeof(uint64_t)));
    AVK_HIP(ctx, hipMalloc((void **)&ctx->sbufs->d_tally, (size_t)AVK_TALLY_STRIDE * sizeof(uint64_t)));
    AVK_HIP(ctx, hipHostMalloc((void **)&ctx->sbufs->h_tally, (size_t)AVK_TALLY_STRIDE * sizeof(uint64_t), hipHostMallocDefault));
    AVK_HIP(ctx, hipMalloc((void **)&ctx->sbufs->d_counters, AVK_N_COUNTERS * sizeof(uint32_t)));
    AVK_HIP(ctx, hipMemset(ctx->sbufs->d_partials, 0, (size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES * sizeof(uint64_t)));
    AVK_HIP(ctx, hipMemset(ctx->sbufs->d_counters, 0, AVK_N_COUNTERS * sizeof(uint32_t)));
    return 0;
}

/* avk_ctx_reserve: pins and allocates what avk_compare_batch needs for a batch of up to n_regions / n_variants, so that a tool can pay for
 * it while it is still reading its inputs (pinning 300 MB and the first device allocations cost 60-100 ms) */
static int stream_reserve(avk_ctx *ctx, uint64_t n_regions, uint64_t n_variants) {
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = stream_bufs_init(ctx);
    if (rc) return rc;
    StreamBufs &sb = *ctx->sbufs;
    rc = grow_pair(ctx, &sb.h_fast, &sb.d_fast, &sb.fast_words, (size_t)(n_regions + 64 * AVK_FAST_CLASSES) * 20); /* a genome's mix needs about 14 words per region; grows when a batch needs more */
    if (!rc) rc = grow_pair(ctx, &sb.h_rout, &sb.d_rout, &sb.rout_words, (size_t)n_regions * 4);
    if (!rc) rc = grow_pair(ctx, &sb.h_vout, &sb.d_vout, &sb.vout_words, (size_t)n_variants + 1);
    if (!rc) rc = grow_pair(ctx, &sb.h_defer, &sb.d_defer, &sb.defer_words, (size_t)n_regions + 64);
    return rc;
}

/* the one-shot path proper; returns AVK_E_STATE + 100 when the batch is not for it (the caller then takes the resident path) */
/* mode 1 = the merge form (avk_optimize_pairs_batch: search A only; out->status and out->ed_h1, which carries `the pair is an exact match`) */
static int compare_batch_stream(avk_ctx *ctx, const avk_region_batch *b, const avk_compare_config *cfg, avk_result_batch *out, uint32_t mode = 0) {
    const uint64_t n = b->n_regions, nv = b->n_variants;
    if (!ctx->lane_kernel || cfg->enable_sequences || cfg->enable_exact_shortcut || cfg->max_branch_factor == 0 || (out->group_metrics && ctx->emit_group_metrics) ||
        n < 32768 || n > 0x7FFFFFFFull || nv > 0x7FFFFFFFull || !ctx->d_ref2b || !ctx->use_packed_reference || ctx->lds_bytes_per_wave == 0 ||
        ctx->ws_bytes_per_wave == 0 || ctx->lds2_overflow_pass)
        return 100;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    const bool timing = getenv("AVK_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    const auto t_begin = now();
    const unsigned nt = avk_host_threads();
    {
        const int rc0 = stream_bufs_init(ctx);
        if (rc0) return rc0;
    }
    StreamBufs &sb = *ctx->sbufs;

    /* ---- 1. classification: which regions the lanes take, their class, their cost key, alt_ed of their calls */
    StreamClass sc;
    sc.cls.assign(n, 0);
    sc.key.assign(n, 0);
    sc.alt_ed.assign(nv ? nv : 1, 0);
    sc.v_off.assign(n + 1, 0);
    const size_t n_contigs = ctx->contig_len.size();
    std::vector<uint64_t> have((size_t)nt * 8 * AVK_FAST_CLASSES, 0); /* regions per class found by every thread (a cache line apart) */
    avk_parallel_for(n, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
        uint64_t *mine = have.data() + (size_t)t * 8 * AVK_FAST_CLASSES;
        for (uint64_t r = lo; r < hi; ++r) {
            const uint32_t tc = b->t_cnt[r], qc = b->q_cnt[r];
            sc.v_off[r + 1] = tc + qc; /* turned into the prefix sum below */
            const uint32_t c = b->contig_idx ? b->contig_idx[r] : 0;
            const uint64_t start = b->start[r], end = b->end[r];
            if (tc > AVK_FAST_MAXV || qc > AVK_FAST_MAXV || tc + qc == 0 || c >= n_contigs || start > end || end > ctx->contig_len[c] || end - start > 255) continue;
            bool ok = true;
            uint64_t ed_sum = 0, grow[2] = {0, 0};
            avk::FastCall calls[2][AVK_FAST_MAXV];
            for (int side = 0; side < 2 && ok; ++side) {
                const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
                const uint32_t cnt = side == 0 ? tc : qc;
                if (off > nv || (uint64_t)cnt > nv - off) {
                    ok = false;
                    break;
                }
                uint64_t last = 0;
                for (uint32_t i = 0; i < cnt && ok; ++i) {
                    const uint64_t v = off + i, pos = b->var_pos[v];
                    const uint32_t l0 = b->a0_len[v], l1 = b->a1_len[v];
                    const uint32_t raw = b->var_raw_space ? b->var_raw_space[v] : (l0 > l1 ? l0 : l1);
                    const uint8_t zy = b->var_zyg[v];
                    ok = l0 >= 1 && l1 >= 1 && l0 <= 255 && l1 <= 32 && raw >= (l0 > l1 ? l0 : l1) && raw <= 0xFFFF && b->var_type[v] < AVK_N_VARIANT_TYPES &&
                         zy >= AVK_ZYG_UNPHASED_HET && zy <= AVK_ZYG_HOM_ALT && pos >= start && pos + l0 <= end && pos >= last &&
                         b->a0_off[v] + l0 <= b->allele_bytes_len && b->a1_off[v] + l1 <= b->allele_bytes_len;
                    if (!ok) break;
                    last = pos;
                    const uint8_t *a0 = b->allele_bytes + b->a0_off[v], *a1 = b->allele_bytes + b->a1_off[v];
                    for (uint32_t j = 0; j < l1 && ok; ++j) ok = a1[j] == 'A' || a1[j] == 'C' || a1[j] == 'G' || a1[j] == 'T';
                    const uint64_t ed = avk::host_edit_distance(a0, l0, a1, l1);
                    ok = ok && ed <= 255;
                    sc.alt_ed[v] = (uint8_t)ed;
                    calls[side][i] = avk::FastCall{(uint32_t)(pos - start), l0, l1, (uint32_t)ed, b->var_type[v], zy, a1};
                    ed_sum += ed;
                    if (l1 > l0) grow[side] += l1 - l0;
                }
            }
            if (!ok || ed_sum > 255) continue;
            const uint64_t L = end - start, g = grow[0] > grow[1] ? grow[0] : grow[1];
            for (int cl = 0; cl < AVK_FAST_CLASSES; ++cl) {
                const AvkFastClass &fc = AVK_FAST_CLASS[cl];
                if (tc <= fc.maxv && qc <= fc.maxv && L + g <= 16ull * fc.W) {
                    sc.cls[r] = (uint8_t)(cl + 1);
                    mine[cl] += 1;
                    const uint8_t ck = avk::fast_cost_key(calls[0], tc, calls[1], qc);
                    if ((int64_t)(ck >> 4) > ctx->lane_max_est) {
                        sc.cls[r] = 0; /* many edits: a whole wavefront's work */
                        mine[cl] -= 1;
                        break;
                    }
                    sc.key[r] = fc.maxv > 2 ? (uint8_t)255 : (uint8_t)(255u - ck); /* ascending = most expensive first; the three-call class keeps the caller's order (avk_pack.h) */
                    break;
                }
            }
        }
    });
    { /* prefix sum of the per-region call counts, in two passes over the threads' ranges (3.5 M additions in a row cost 3 ms of a 40 ms call) */
        std::vector<uint64_t> part((size_t)nt + 1, 0);
        avk_parallel_for(n, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
            uint64_t sum = 0;
            for (uint64_t r = lo; r < hi; ++r) sum += sc.v_off[r + 1];
            part[t + 1] = sum;
        });
        for (unsigned t = 0; t < nt; ++t) part[t + 1] += part[t];
        avk_parallel_for(n, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
            uint32_t run = (uint32_t)part[t];
            for (uint64_t r = lo; r < hi; ++r) {
                run += sc.v_off[r + 1];
                sc.v_off[r + 1] = run;
            }
        });
    }
    if (sc.v_off[n] > 0x7FFFFFFFull) return 100;
    { /* a class too small for a launch of its own (plan_work_order's rule, option lane_min_regions) joins the general part */
        static const uint64_t scale[AVK_FAST_CLASSES] = {1, 1, 16, 16, 2};
        bool drop[AVK_FAST_CLASSES], any = false;
        for (int cl = 0; cl < AVK_FAST_CLASSES; ++cl) {
            uint64_t c = 0;
            for (unsigned t = 0; t < nt; ++t) c += have[(size_t)t * 8 * AVK_FAST_CLASSES + cl];
            /* the three-call class stays with the general part here: that part is solved by the resident path (which has the class, its node
             * budget and the launch behind it) from the start of the call, beside the packing, while regions handed back by THIS path's lanes
             * wait for a second call at the end (measured: 42 ms per whole-genome call this way, 50 ms with the class in the one-shot launches) */
            drop[cl] = c > 0 && (c < (uint64_t)ctx->lane_min_regions * scale[cl] || (int64_t)AVK_FAST_CLASS[cl].maxv > ctx->lane_max_calls || AVK_FAST_CLASS[cl].maxv > 2);
            any = any || drop[cl];
        }
        if (any)
            avk_parallel_for(n, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
                for (uint64_t r = lo; r < hi; ++r)
                    if (sc.cls[r] && drop[sc.cls[r] - 1]) sc.cls[r] = 0, sc.key[r] = 0;
            });
    }
    const uint64_t nv_dev = sc.v_off[n];
    /* counting sort, stable, on the host threads: per class, by key (most expensive first); every thread owns a contiguous range of regions */
    enum { KEYS = 256, NB = (AVK_FAST_CLASSES + 1) * KEYS };
    const unsigned st = nt > 1 && n >= 16384 ? nt : 1;
    std::vector<uint64_t> hist((size_t)st * NB, 0);
    avk_parallel_for(n, st, [&](unsigned t, uint64_t lo, uint64_t hi) {
        uint64_t *h = hist.data() + (size_t)t * NB;
        for (uint64_t r = lo; r < hi; ++r) h[(uint32_t)sc.cls[r] * KEYS + sc.key[r]] += 1;
    });
    uint64_t cnt[NB + 1];
    {
        uint64_t run = 0;
        for (int k = 0; k < NB; ++k) {
            cnt[k] = run;
            for (unsigned t = 0; t < st; ++t) {
                const uint64_t c = hist[(size_t)t * NB + k];
                hist[(size_t)t * NB + k] = run; /* where thread t writes its first region of bucket k */
                run += c;
            }
        }
        cnt[NB] = run;
    }
    const uint64_t n_general = cnt[KEYS];
    uint64_t class_lo[AVK_FAST_CLASSES + 1];
    for (int cl = 0; cl <= AVK_FAST_CLASSES; ++cl) class_lo[cl] = cnt[(uint32_t)(cl + 1) * KEYS]; /* start of cls == cl + 1 */
    uint64_t n_heavy[AVK_FAST_CLASSES]; /* regions with estimated edits (cost key >> 4 != 0, i.e. sort key below 240): the head of the class */
    for (int cl = 0; cl < AVK_FAST_CLASSES; ++cl) n_heavy[cl] = cnt[(uint32_t)(cl + 1) * KEYS + 240] - cnt[(uint32_t)(cl + 1) * KEYS];
    std::vector<uint32_t> order(n); /* [general | class 0 | class 1 | ..] */
    avk_parallel_for(n, st, [&](unsigned t, uint64_t lo, uint64_t hi) {
        uint64_t *at = hist.data() + (size_t)t * NB;
        for (uint64_t r = lo; r < hi; ++r) order[at[(uint32_t)sc.cls[r] * KEYS + sc.key[r]]++] = (uint32_t)r;
    });
    const uint64_t n_fast = n - n_general;
    if (n_fast < 16384) return 100;
    uint32_t tile_base[AVK_FAST_CLASSES], n_tiles[AVK_FAST_CLASSES], tiles = 0;
    uint64_t n_class[AVK_FAST_CLASSES], word_base[AVK_FAST_CLASSES + 1], words = 0;
    /* the classes with the most calls per side come first in the record array: they are written and copied first, and their launches — the
     * long ones — start while the one-call classes are still being written */
    for (int cl = AVK_FAST_CLASSES - 1; cl >= 0; --cl) {
        n_class[cl] = class_lo[cl + 1] - class_lo[cl];
        tile_base[cl] = tiles;
        word_base[cl] = words;
        n_tiles[cl] = (uint32_t)((n_class[cl] + 63) / 64);
        tiles += n_tiles[cl];
        words += (uint64_t)n_tiles[cl] * AVK_FAST_WORDS_OF(AVK_FAST_CLASS[cl].maxv) * 64u;
    }
    word_base[AVK_FAST_CLASSES] = words;
    auto word_of_tile = [&](uint32_t tile) { /* first word of a tile (tile == tiles: the end) */
        for (int k = 0; k < AVK_FAST_CLASSES; ++k)
            if (tile >= tile_base[k] && tile < tile_base[k] + n_tiles[k])
                return word_base[k] + (uint64_t)(tile - tile_base[k]) * AVK_FAST_WORDS_OF(AVK_FAST_CLASS[k].maxv) * 64u;
        return words;
    };
    const auto t_class = now();

    /* ---- 2. the general part goes first, through the resident path: its kernels run while the fast records are written */
    avk_dev_batch *dbG = nullptr;
    std::vector<uint64_t> g_rid, g_start, g_end, g_toff, g_qoff;
    std::vector<uint32_t> g_cidx, g_tcnt, g_qcnt;
    avk_region_batch G = *b;
    auto gather = [&](const uint32_t *idx, uint64_t m) {
        g_rid.resize(m), g_start.resize(m), g_end.resize(m), g_toff.resize(m), g_qoff.resize(m), g_cidx.resize(m), g_tcnt.resize(m), g_qcnt.resize(m);
        for (uint64_t k = 0; k < m; ++k) {
            const uint32_t r = idx[k];
            g_rid[k] = b->region_id ? b->region_id[r] : r;
            g_cidx[k] = b->contig_idx ? b->contig_idx[r] : 0;
            g_start[k] = b->start[r], g_end[k] = b->end[r], g_toff[k] = b->t_off[r], g_tcnt[k] = b->t_cnt[r], g_qoff[k] = b->q_off[r], g_qcnt[k] = b->q_cnt[r];
        }
        G.n_regions = m;
        G.region_id = g_rid.data(), G.contig_idx = g_cidx.data(), G.start = g_start.data(), G.end = g_end.data();
        G.t_off = g_toff.data(), G.t_cnt = g_tcnt.data(), G.q_off = g_qoff.data(), G.q_cnt = g_qcnt.data();
    };
    /* results of a gathered batch: fetched into temporaries (the per-variant outputs go straight to the caller's arrays, which the gathered
     * batch shares), then scattered into the caller's per-region arrays */
    struct Gathered {
        std::vector<int32_t> st;
        std::vector<uint32_t> e1, e2, no;
        std::vector<uint16_t> tp;
        std::vector<uint64_t> tl;
    };
    auto fetch_gathered = [&](avk_dev_batch *db, uint64_t m, Gathered &g) -> int {
        g.st.resize(m + 1), g.e1.resize(m + 1), g.e2.resize(m + 1), g.no.resize(m + 1), g.tp.resize(m + 1), g.tl.assign(AVK_TALLY_LEN, 0);
        if (mode == 1) { /* the pair form has per-region records only (as avk_optimize_pairs_batch reads them) */
            std::vector<uint32_t> rout(m * 4 + 4);
            hipError_t e = hipMemcpyAsync(rout.data(), db->d_region_out, m * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return fail(ctx, AVK_E_HIP, "pairs download failed: %s", hipGetErrorString(e));
            for (uint64_t k = 0; k < m; ++k) g.st[k] = (int32_t)rout[4 * k], g.e1[k] = rout[4 * k + 1], g.e2[k] = 0, g.no[k] = 0, g.tp[k] = 0;
            return 0;
        }
        avk_result_batch o;
        memset(&o, 0, sizeof(o));
        o.status = g.st.data(), o.ed_h1 = g.e1.data(), o.ed_h2 = g.e2.data(), o.n_optima = g.no.data(), o.type_present = g.tp.data(), o.tally = g.tl.data();
        o.var_expected = out->var_expected, o.var_observed = out->var_observed, o.var_class = out->var_class, o.var_zyg = out->var_zyg;
        return avk_results_download(ctx, db, &o);
    };
    auto scatter_gathered = [&](const Gathered &g, const uint32_t *idx, uint64_t m, uint64_t *tally_sum) {
        for (uint64_t k = 0; k < m; ++k) {
            const uint32_t r = idx[k];
            out->status[r] = g.st[k];
            if (out->ed_h1) out->ed_h1[r] = g.e1[k];
            if (out->ed_h2) out->ed_h2[r] = g.e2[k];
            if (out->n_optima) out->n_optima[r] = g.no[k];
            if (out->type_present) out->type_present[r] = g.tp[k];
        }
        for (int i = 0; i < AVK_TALLY_LEN; ++i) tally_sum[i] += g.tl[i];
    };
    const int64_t keep_gm = ctx->emit_group_metrics;
    ctx->emit_group_metrics = 0;
    int rc = 0, rc_general = 0;
    Gathered resG, resD;
    std::thread general_thread; /* packs, uploads and launches the general part while this thread writes the fast records */
    if (n_general) {
        gather(order.data(), n_general);
        general_thread = std::thread([&] {
            (void)hipSetDevice(ctx->device);
            rc_general = upload_internal(ctx, &G, mode == 1, &dbG);
            if (!rc_general) rc_general = run_internal(ctx, dbG, cfg, nullptr, mode);
            if (!rc_general) rc_general = fetch_gathered(dbG, n_general, resG); /* waits for the general kernels on this thread */
            if (dbG) avk_batch_free(ctx, dbG);
            dbG = nullptr;
        });
    }
    const auto t_general = now();

    /* ---- 3. fast records: written into pinned memory tile range by tile range, each range copied while the next is written */
    if (!rc) rc = grow_pair(ctx, &sb.h_fast, &sb.d_fast, &sb.fast_words, (size_t)words + 64);
    if (!rc) rc = grow_pair(ctx, &sb.h_rout, &sb.d_rout, &sb.rout_words, (size_t)n * 4);
    if (!rc) rc = grow_pair(ctx, &sb.h_vout, &sb.d_vout, &sb.vout_words, (size_t)nv_dev + 1);
    if (!rc) rc = grow_pair(ctx, &sb.h_defer, &sb.d_defer, &sb.defer_words, (size_t)n_fast + 64);
    if (rc) {
        if (general_thread.joinable()) general_thread.join();
        if (dbG) avk_batch_free(ctx, dbG);
        ctx->emit_group_metrics = keep_gm;
        return rc;
    }
    const uint32_t *fast_order = order.data() + n_general; /* position in here = what a lane reports when it hands a region back */
    auto tile_class = [&](uint32_t tile) {
        int cl = 0;
        for (int k = 0; k < AVK_FAST_CLASSES; ++k)
            if (tile >= tile_base[k] && tile < tile_base[k] + n_tiles[k]) cl = k;
        return cl;
    };
    /* the host threads take pieces of 64 tiles from one counter, in tile order; this thread queues the copy of every range of 1024 tiles
     * as soon as its pieces are written, so the copies run beside the writing */
    const uint32_t piece = 64, chunk_tiles = 1024;
    const uint32_t n_pieces = (tiles + piece - 1) / piece, n_chunks = (tiles + chunk_tiles - 1) / chunk_tiles;
    std::atomic<uint32_t> next_piece(0);
    std::vector<std::atomic<uint32_t>> chunk_done(n_chunks);
    for (auto &x : chunk_done) x.store(0);
    auto pack_worker = [&] {
        for (;;) {
            const uint32_t p = next_piece.fetch_add(1);
            if (p >= n_pieces) break;
            const uint32_t t0 = p * piece, t1 = t0 + piece < tiles ? t0 + piece : tiles;
            for (uint32_t tile = t0; tile < t1; ++tile)
                for (uint32_t lane = 0; lane < 64; ++lane) {
                const int cl = tile_class(tile);
                const uint32_t maxv = AVK_FAST_CLASS[cl].maxv, rw = AVK_FAST_WORDS_OF(maxv);
                uint32_t *T = sb.h_fast + word_base[cl] + (size_t)(tile - tile_base[cl]) * rw * 64u + lane;
                const uint64_t k = (uint64_t)(tile - tile_base[cl]) * 64u + lane;
                if (k >= n_class[cl]) {
                    for (uint32_t w = 0; w < rw; ++w) T[w * 64] = w == 1 ? 0xFFFFFFFFu : 0u;
                    continue;
                }
                const uint32_t r = order[class_lo[cl] + k];
                const uint32_t tc = b->t_cnt[r], qc = b->q_cnt[r];
                const uint64_t start = b->start[r];
                const uint64_t ref_off = ctx->contig_base[b->contig_idx ? b->contig_idx[r] : 0] + start;
                uint32_t slot_pos[2 * AVK_FAST_MAXV] = {0};
                for (uint32_t s = 0; s < 2 * AVK_FAST_MAXV; ++s) { /* s = side * AVK_FAST_MAXV + the call's index on its side */
                    const uint32_t side = s / AVK_FAST_MAXV, j = s % AVK_FAST_MAXV;
                    if (j >= maxv) continue; /* the class's records have no such slot */
                    const bool on = j < (side ? qc : tc);
                    uint32_t *V = T + (AVK_FAST_HDR + 4 * (side * maxv + j)) * 64;
                    if (!on) {
                        V[0] = V[64] = V[128] = V[192] = 0;
                        continue;
                    }
                    const uint64_t v = (side ? b->q_off[r] : b->t_off[r]) + j;
                    const uint32_t l0 = b->a0_len[v], l1 = b->a1_len[v];
                    const uint32_t raw = b->var_raw_space ? b->var_raw_space[v] : (l0 > l1 ? l0 : l1);
                    const uint8_t *a1 = b->allele_bytes + b->a1_off[v];
                    slot_pos[s] = (uint32_t)(b->var_pos[v] - start);
                    V[0] = slot_pos[s] | (l0 << 8) | (l1 << 16) | ((uint32_t)(b->var_type[v] & 0xFu) << 24) | ((uint32_t)(b->var_zyg[v] & 7u) << 28);
                    V[64] = (uint32_t)sc.alt_ed[v] | (raw << 8);
                    V[128] = avk::pack_bases_2bit(a1, l1);
                    V[192] = l1 > 16 ? avk::pack_bases_2bit(a1 + 16, l1 - 16) : 0u;
                }
                /* order_variants (query_optimizer.rs:372-381): stable merge by position, truth first on ties; bit d = depth d takes a query call */
                uint32_t ord = 0, i = 0, j = 0, d = 0;
                while (i < tc || j < qc) {
                    const bool take_t = j >= qc || (i < tc && slot_pos[i] <= slot_pos[AVK_FAST_MAXV + j]);
                    if (take_t) i++;
                    else j++, ord |= 1u << d;
                    d++;
                }
                T[0] = (uint32_t)(ref_off >> 4);
                T[64] = (uint32_t)(ref_off & 15u) | ((uint32_t)(b->end[r] - start) << 4) | (tc << 12) | (qc << 14) | (ord << 16);
                T[128] = sc.v_off[r];
                T[192] = r;
                }
            chunk_done[t0 / chunk_tiles].fetch_add(1, std::memory_order_release);
        }
    };
    hipError_t herr = hipSuccess;
    /* ---- 4. the lane launches, class by class as soon as a class's records are on their way: two-call classes, three-call class and
     * one-call classes on a lane stream each (as in run_internal), behind an event on the copy stream */
    if (!ctx->lane_attr_set) {
        herr = hipFuncSetAttribute((const void *)avk_lane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ctx->lane_attr_set = herr == hipSuccess;
    }
    AvkKernelArgs f;
    memset(&f, 0, sizeof(f));
    f.ref_bytes = ctx->d_ref;
    f.ref_2bit = ctx->d_ref2b;
    f.ref_exc = ctx->d_refexc;
    f.n_regions = (uint32_t)n;
    f.max_branch_factor = cfg->max_branch_factor;
    f.mode = mode;
    f.region_out = sb.d_rout;
    f.var_out = sb.d_vout;
    f.tally = sb.d_partials;
    f.overflow_list = sb.d_defer;
    f.overflow_count = sb.d_counters + 1024;
    bool stream_used[3] = {false, false, false}; /* lane_stream (two calls), lane_stream2 (one call), lane_stream3 (three calls) */
    auto launch_class = [&](int cl) { /* on the thread that queues the copies; everything of the class has been queued for copying */
        if (!n_tiles[cl] || herr != hipSuccess) return;
        const AvkFastClass &fcl = AVK_FAST_CLASS[cl];
        avk::lane::LaneArgs la;
        la.recs = sb.d_fast + word_base[cl];
        la.rec_words = AVK_FAST_WORDS_OF(fcl.maxv);
        la.n_tiles = n_tiles[cl];
        la.tile_counter = sb.d_counters + 1220 + cl;
        la.W = fcl.W;
        la.nm = 1u << fcl.maxv;
        la.ed_max = fcl.ed_max;
        la.qcap = fcl.qcap;
        la.gen_base = (uint32_t)(class_lo[cl] - n_general);
        la.lanes_log2 = lane_width_log2(ctx, fcl.maxv);
        la.max_nodes = fcl.maxv > 2 ? (uint32_t)ctx->lane_node_cap : 250u;
        la.max_ed_c = (uint32_t)ctx->lane_metrics_ed_cap;
        uint32_t grid = 0;
        const size_t lds = lane_launch_geometry(ctx, la, &grid);
        if (!lds) {
            herr = hipErrorInvalidValue;
            return;
        }
        const int si = fcl.maxv == 2 ? 0 : (fcl.maxv == 1 ? 1 : 2);
        hipStream_t ls = si == 0 ? ctx->lane_stream : (si == 1 ? ctx->lane_stream2 : ctx->lane_stream3);
        if (si == 1 && n_chunks <= 1) ls = sb.copy_stream; /* a small batch: behind its one copy, no event hop (a chr20-sized call is 1.3 ms in all) */
        else {
            herr = hipEventRecord(sb.ev_copied, sb.copy_stream); /* the class's last copy is ahead of this record */
            if (herr == hipSuccess) herr = hipStreamWaitEvent(ls, sb.ev_copied, 0);
            stream_used[si] = true;
        }
        const uint32_t head_tiles = ctx->lane_head_width ? (uint32_t)((n_heavy[cl] + 63u) / 64u) : 0u; /* as in run_internal */
        if (herr == hipSuccess && head_tiles > 0 && head_tiles < la.n_tiles && (uint32_t)ctx->lane_head_width < (1u << la.lanes_log2)) {
            avk::lane::LaneArgs hd = la;
            hd.n_tiles = head_tiles;
            hd.tile_counter = sb.d_counters + 1230 + cl;
            hd.lanes_log2 = head_width_log2(ctx);
            uint32_t hgrid = 0;
            const size_t hlds = lane_launch_geometry(ctx, hd, &hgrid);
            hipLaunchKernelGGL(avk_lane_kernel, dim3(hgrid), dim3(64), hlds, ls, f, hd);
            herr = hipGetLastError();
            la.recs += (size_t)head_tiles * la.rec_words * 64u;
            la.n_tiles -= head_tiles;
            la.gen_base += head_tiles * 64u;
            if (grid > la.n_tiles) grid = la.n_tiles;
        }
        if (herr == hipSuccess) {
            hipLaunchKernelGGL(avk_lane_kernel, dim3(grid), dim3(64), lds, ls, f, la);
            herr = hipGetLastError();
        }
    };
    int next_class = AVK_FAST_CLASSES - 1; /* classes are complete in the order of the record array */
    auto copy_loop = [&] {
        for (uint32_t ch = 0; ch < n_chunks && herr == hipSuccess; ++ch) {
            const uint32_t t0 = ch * chunk_tiles, t1 = t0 + chunk_tiles < tiles ? t0 + chunk_tiles : tiles;
            const uint32_t want = (t1 - t0 + piece - 1) / piece;
            while (chunk_done[ch].load(std::memory_order_acquire) < want) std::this_thread::sleep_for(std::chrono::microseconds(20));
            const size_t w0 = (size_t)word_of_tile(t0), w1 = (size_t)word_of_tile(t1);
            herr = hipMemcpyAsync(sb.d_fast + w0, sb.h_fast + w0, (w1 - w0) * sizeof(uint32_t), hipMemcpyHostToDevice, sb.copy_stream);
            while (herr == hipSuccess && next_class >= 0 && tile_base[next_class] + n_tiles[next_class] <= t1) launch_class(next_class--);
        }
    };
    AvkPool::get().run(nt + 1, [&](unsigned t) { /* this thread queues the copies and the launches, the pool writes the records */
        if (t == 0) copy_loop();
        else pack_worker();
    });
    while (herr == hipSuccess && next_class >= 0) launch_class(next_class--);
    const auto t_packed = now();
    {
        hipStream_t lss[3] = {ctx->lane_stream, ctx->lane_stream2, ctx->lane_stream3};
        hipEvent_t evs[3] = {sb.ev_lane_done, sb.ev_lane_done2, sb.ev_lane_done3};
        for (int si = 0; si < 3 && herr == hipSuccess; ++si) {
            if (!stream_used[si]) continue;
            herr = hipEventRecord(evs[si], lss[si]);
            if (herr == hipSuccess) herr = hipStreamWaitEvent(sb.copy_stream, evs[si], 0);
        }
    }
    /* the handed-back list first (it decides the second call, which starts beside the download of the lane results) */
    if (herr == hipSuccess) herr = hipMemcpyAsync(sb.h_defer, sb.d_counters + 1024, sizeof(uint32_t), hipMemcpyDeviceToHost, sb.copy_stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(sb.copy_stream);
    const uint32_t n_defer = herr == hipSuccess ? sb.h_defer[0] : 0;
    if (herr == hipSuccess && n_defer) {
        herr = hipMemcpyAsync(sb.h_defer, sb.d_defer, (size_t)n_defer * sizeof(uint32_t), hipMemcpyDeviceToHost, sb.copy_stream);
        if (herr == hipSuccess) herr = hipStreamSynchronize(sb.copy_stream);
    }
    if (general_thread.joinable()) general_thread.join();
    if (!rc) rc = rc_general;
    std::vector<uint32_t> defer_idx(herr == hipSuccess ? n_defer : 0);
    for (size_t k = 0; k < defer_idx.size(); ++k) {
        defer_idx[k] = fast_order[sb.h_defer[k]];
        sc.cls[defer_idx[k]] = 0; /* not a lane result after all */
    }
    int rc_tail = 0;
    std::thread tail_thread;
    if (!rc && !defer_idx.empty())
        tail_thread = std::thread([&] { /* wave-per-region kernels only: the lanes gave these back */
            (void)hipSetDevice(ctx->device);
            std::sort(defer_idx.begin(), defer_idx.end());
            gather(defer_idx.data(), n_defer);
            avk_dev_batch *dbD = nullptr;
            const int64_t keep_lane = ctx->lane_kernel;
            ctx->lane_kernel = 0;
            rc_tail = upload_internal(ctx, &G, mode == 1, &dbD);
            if (!rc_tail) rc_tail = run_internal(ctx, dbD, cfg, nullptr, mode);
            ctx->lane_kernel = keep_lane;
            if (!rc_tail) rc_tail = fetch_gathered(dbD, n_defer, resD);
            if (dbD) avk_batch_free(ctx, dbD);
        });
    if (herr == hipSuccess) {
        hipLaunchKernelGGL(avk_tally_reduce, dim3((AVK_TALLY_STRIDE + 63) / 64), dim3(64), 0, sb.copy_stream, sb.d_partials, sb.d_tally, (uint64_t *)nullptr, sb.d_counters,
                           (unsigned)AVK_N_COUNTERS, 0u);
        herr = hipGetLastError();
    }
    if (herr == hipSuccess) herr = hipMemcpyAsync(sb.h_tally, sb.d_tally, (size_t)AVK_TALLY_STRIDE * sizeof(uint64_t), hipMemcpyDeviceToHost, sb.copy_stream);
    if (herr == hipSuccess) herr = hipMemcpyAsync(sb.h_rout, sb.d_rout, (size_t)n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, sb.copy_stream);
    if (herr == hipSuccess && nv_dev && mode == 0) herr = hipMemcpyAsync(sb.h_vout, sb.d_vout, (size_t)nv_dev * sizeof(uint32_t), hipMemcpyDeviceToHost, sb.copy_stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(sb.copy_stream);
    const auto t_lanes = now();
    if (herr != hipSuccess) {
        if (tail_thread.joinable()) tail_thread.join();
        ctx->emit_group_metrics = keep_gm;
        return fail(ctx, AVK_E_HIP, "one-shot compare failed: %s", hipGetErrorString(herr));
    }

    /* ---- 5. results: lane part unpacked on the host threads beside the second call; general and handed-back parts scattered after */
    std::vector<uint64_t> tally_sum(AVK_TALLY_LEN, 0);
    for (int i = 0; i < AVK_TALLY_LEN; ++i) tally_sum[i] = sb.h_tally[i];
    if (n_general && !rc) scatter_gathered(resG, order.data(), n_general, tally_sum.data());
    const auto t_defer_done = now();
    avk_parallel_for(n, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; ++r) {
            if (!sc.cls[r]) continue;
            const uint32_t *w = sb.h_rout + 4 * r;
            out->status[r] = (int32_t)w[0];
            if (out->ed_h1) out->ed_h1[r] = w[1];
            if (out->ed_h2) out->ed_h2[r] = w[2];
            if (out->n_optima) out->n_optima[r] = w[3] & 0xFFFFu;
            if (out->type_present) out->type_present[r] = (uint16_t)(w[3] >> 16);
            if (mode != 0) continue; /* the pair form has no per-call outputs */
            const uint32_t tc = b->t_cnt[r], qc = b->q_cnt[r];
            const uint32_t *vw = sb.h_vout + sc.v_off[r];
            for (uint32_t k = 0; k < tc + qc; ++k) {
                const uint64_t hv = k < tc ? b->t_off[r] + k : b->q_off[r] + (k - tc);
                const uint32_t x = vw[k];
                if (out->var_expected) out->var_expected[hv] = (uint8_t)(x & 0xFF);
                if (out->var_observed) out->var_observed[hv] = (uint8_t)((x >> 8) & 0xFF);
                if (out->var_class) out->var_class[hv] = (uint8_t)((x >> 16) & 0xFF);
                if (out->var_zyg) out->var_zyg[hv] = (uint8_t)(x >> 24);
            }
        }
    });
    const auto t_unpacked = now();
    if (tail_thread.joinable()) tail_thread.join();
    if (!rc && !rc_tail && n_defer) scatter_gathered(resD, defer_idx.data(), n_defer, tally_sum.data());
    ctx->emit_group_metrics = keep_gm;
    if (rc || rc_tail) return rc ? rc : rc_tail;
    if (out->tally) memcpy(out->tally, tally_sum.data(), AVK_TALLY_LEN * sizeof(uint64_t));
    ctx->last_lane_solved = sb.h_tally[AVK_TALLY_LANE_SOLVED];
    if (timing)
        fprintf(stderr,
                "avk one-shot compare: %llu regions (%llu general, %llu fast in %u tiles, %u handed back); classify %.2f ms, general part started %.2f ms, fast records + H2D "
                "%.2f ms, lane kernels + D2H %.2f ms, unpack %.2f ms beside general results + handed-back call (%.2f ms more); total %.2f ms on %u host threads\n",
                (unsigned long long)n, (unsigned long long)n_general, (unsigned long long)n_fast, tiles, n_defer, ms(t_begin, t_class), ms(t_class, t_general),
                ms(t_general, t_packed), ms(t_packed, t_lanes), ms(t_defer_done, t_unpacked), ms(t_unpacked, now()), ms(t_begin, now()), nt);
    return 0;
}
