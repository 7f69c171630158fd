/*
 * wave_emu.cpp — kernel-logic emulator for the GPU-less dev container.  TEST INFRASTRUCTURE.
 *
 * Compiles the SAME solver source the HIP library is built from (aardvark_amd/csrc/avk_solver.inl)
 * with AVK_EMU and executes each 64-lane wavefront as 64 cooperative fibers on one OS thread.
 * Every wave primitive (ballot, shuffle, reductions, wv_sync, wv_uni) is a rendezvous of the 64
 * fibers; the call site of each rendezvous is compared across lanes, so divergent use of a wave
 * primitive (which would hang or mis-ballot on the GPU) aborts the test.  Memory is plain host
 * memory: the "LDS slice" and the HBM workspaces are heap buffers.
 *
 * Used by tests/test_emu_parity.py to fuzz the kernel logic against the oracle without a GPU.
 * Never linked into libaardvark_amd.so; not a fallback path.
 */
#define AVK_EMU 1
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../aardvark_amd/csrc/avk_wave.h"

/* ---- fibers ------------------------------------------------------------------------------ */
extern "C" void avk_emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl avk_emu_switch
.type avk_emu_switch,@function
avk_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size avk_emu_switch,.-avk_emu_switch
)");

namespace avk_emu {

struct Wave {
    void *sp[64];
    void *main_sp;
    char *stacks;
    size_t stack_bytes;
    int cur;
    uint32_t gen[64];
    bool done[64];
    uint64_t slots[2][64];
    uint32_t sites[2][64];
    void (*fn)(void *, int);
    void *arg;
};
static thread_local Wave *t_wave = nullptr;

int lane() { return t_wave->cur; }
void yield() { std::this_thread::yield(); }

static void switch_to(Wave *w, int from, int to) {
    w->cur = to;
    avk_emu_switch(&w->sp[from], w->sp[to]);
}

const uint64_t *gather(uint64_t v, uint32_t site) {
    Wave *w = t_wave;
    const int me = w->cur;
    const uint32_t g = w->gen[me]++;
    w->slots[g & 1][me] = v;
    w->sites[g & 1][me] = site;
    switch_to(w, me, (me + 1) & 63);
    /* back: every lane has deposited generation g */
    const uint32_t *s = w->sites[g & 1];
    for (int i = 0; i < 64; ++i) {
        if (s[i] != site || w->gen[i] < g + 1) {
            fprintf(stderr, "avk_emu: DIVERGENT wave primitive: lane %d at site %u, lane %d at site %u (gen %u/%u)\n", me, site & 0x7FFFFFFFu, i,
                    s[i] & 0x7FFFFFFFu, g, w->gen[i]);
            abort();
        }
    }
    if (site & 0x80000000u) { /* wv_uni: the value must already be uniform */
        const uint64_t *vals = w->slots[g & 1];
        for (int i = 1; i < 64; ++i)
            if (vals[i] != vals[0]) {
                fprintf(stderr, "avk_emu: wv_uni on a NON-UNIFORM value at site %u: lane0=%llu lane%d=%llu\n", site & 0x7FFFFFFFu,
                        (unsigned long long)vals[0], i, (unsigned long long)vals[i]);
                abort();
            }
    }
    return w->slots[g & 1];
}

static void fiber_entry() {
    Wave *w = t_wave;
    const int me = w->cur;
    w->fn(w->arg, me);
    w->done[me] = true;
    /* all lanes must have passed the same number of rendezvous */
    for (int i = 0; i < 64; ++i)
        if (w->gen[i] != w->gen[me]) {
            fprintf(stderr, "avk_emu: lane %d finished after %u rendezvous but lane %d is at %u\n", me, w->gen[me], i, w->gen[i]);
            abort();
        }
    if (me == 63) {
        w->cur = -1;
        void *dummy;
        avk_emu_switch(&dummy, w->main_sp);
    } else {
        void *dummy;
        w->cur = me + 1;
        avk_emu_switch(&dummy, w->sp[me + 1]);
    }
    abort(); /* unreachable */
}

/* runs fn(arg, lane) on 64 lanes as one wavefront */
static void run_wave(Wave *w, void (*fn)(void *, int), void *arg) {
    w->fn = fn;
    w->arg = arg;
    for (int i = 0; i < 64; ++i) {
        w->gen[i] = 0;
        w->done[i] = false;
        char *top = w->stacks + (size_t)(i + 1) * w->stack_bytes;
        uintptr_t t = ((uintptr_t)top & ~(uintptr_t)15) - 8; /* rsp % 16 == 8 at function entry */
        void **sp = (void **)t;
        *--sp = (void *)&fiber_entry; /* return address popped by `ret` */
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        w->sp[i] = (void *)sp;
    }
    t_wave = w;
    w->cur = 0;
    avk_emu_switch(&w->main_sp, w->sp[0]);
    t_wave = nullptr;
}

} // namespace avk_emu

#include "../../aardvark_amd/csrc/avk_pack.h"
#include "../../aardvark_amd/csrc/avk_solver.inl"
#ifdef AVK_LANE_STATS
namespace avk { namespace lane { uint64_t g_lane_stats[32]; int g_lane_phase; uint32_t *g_lane_work; } }
#endif
#include "../../aardvark_amd/csrc/avk_lane.inl"
#include "../../aardvark_amd/csrc/avk_dwfa_script.inl"

namespace {

int g_lane_kernel = 1; /* emu_set_lane_kernel: small regions through the lane-per-region code (avk_lane.inl), as run_internal does */
uint64_t g_lane_solved = 0;
uint32_t g_lane_width_log2[3] = {6, 6, 4};
uint32_t g_lane_node_cap = 32;
uint32_t g_lane_head_width = 16; /* run_internal's option lane_head_width */ /* emu_set_lane_width: records a wave takes at a time in the one-call / two-call classes */

struct LaneTask {
    const AvkKernelArgs *args;
    const avk::lane::LaneArgs *la;
    uint32_t wave_id;
    uint32_t *lds, *tally;
    uint32_t n_ok[64], n_err[64];
};
void lane_kernel_main(void *p, int lane) {
    LaneTask *t = (LaneTask *)p;
    uint32_t ok = 0, err = 0;
    avk::lane::lane_worker(*t->args, *t->la, t->wave_id, t->lds, t->tally, ok, err);
    t->n_ok[lane] = ok;
    t->n_err[lane] = err;
}

struct WaveTask {
    const AvkKernelArgs *args;
    uint32_t wave_id;
    uint8_t *lds;
    uint8_t *wg_lds = nullptr; /* the workgroup's whole LDS when the in-workgroup escalation is on */
    uint32_t wave_in_wg = 0, n_wg_waves = 0;
};

void lane_main(void *p, int /*lane*/) {
    WaveTask *t = (WaveTask *)p;
    if (t->lds) avk::region_worker<true>(*t->args, t->wave_id, t->lds);
    else avk::region_worker<false>(*t->args, t->wave_id, nullptr);
}

} // namespace

extern "C" {

/* Mirrors avk_compare_batch on emulated wavefronts.  lds_bytes / lds2_bytes / ws_bytes / big_ws_bytes
 * are the per-wave workspace sizes of the four tiers (0 disables a tier), *_ed_cap the wavefront caps
 * of the LDS tiers, n_waves the number of persistent waves, threads the OS threads running them.
 * tier_counts[5] receives how many regions each tier finished, then the capacity failures. */
static int emu_run(uint32_t mode, const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
                      const avk_compare_config *cfg, avk_result_batch *out, uint64_t lds_bytes, uint32_t lds_ed_cap, uint64_t lds2_bytes,
                      uint32_t lds2_ed_cap, uint64_t ws_bytes, uint64_t big_ws_bytes, uint32_t n_waves, int threads, uint64_t *tier_counts,
                      uint32_t solo_min_variants, uint32_t lds2_overflow_pass, uint32_t lds_escalation) {
    std::vector<uint64_t> base(n_contigs), lens(n_contigs);
    uint64_t total = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        base[c] = total;
        lens[c] = ref_lens[c];
        total += ref_lens[c];
    }
    std::vector<uint8_t> refcat(total + 1);
    for (uint32_t c = 0; c < n_contigs; ++c) memcpy(refcat.data() + base[c], refs[c], ref_lens[c]);

    avk::PackedBatch pb;
    std::string err;
    const bool want_seq = cfg->enable_sequences && out->seq_bytes && out->seq_len && out->seq_off && out->seq_stride;
    int rc = avk::pack_batch(batch, base, lens, want_seq ? out->seq_off : nullptr, want_seq ? out->seq_stride : nullptr, &pb, &err);
    if (rc) {
        fprintf(stderr, "emu pack error: %s\n", err.c_str());
        return rc;
    }
    const uint64_t n = batch->n_regions, nv = pb.variants.size();
    std::vector<uint32_t> rout(n * 4 + 4, 0), vout(nv + 1, 0);
    std::vector<uint32_t> gm(out->group_metrics ? n * AVK_N_GROUPS * AVK_N_FIELDS : 0);
    std::vector<uint64_t> partials((size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES, 0), tally(AVK_TALLY_STRIDE, 0);
    std::vector<uint32_t> lists[4] = {std::vector<uint32_t>(n + 1), std::vector<uint32_t>(n + 1), std::vector<uint32_t>(n + 1), std::vector<uint32_t>(n + 1)};
    std::vector<uint32_t> counters_v(1280, 0);
    uint32_t *counters = counters_v.data();

    AvkKernelArgs a;
    memset(&a, 0, sizeof(a));
    a.regions = pb.regions.data();
    a.blob = pb.blob.data();
    a.ref_bytes = refcat.data();
    /* packed copy, as avk_pack_reference builds it on the device */
    const uint64_t n_words = (total + 15) >> 4;
    std::vector<uint32_t> ref2b(n_words + 80, 0), refexc((n_words >> 5) + 8, 0);
    for (uint64_t p = 0; p < total; ++p) {
        const uint8_t ch = refcat[p];
        uint32_t code = 0;
        if (ch == 'A') code = 0;
        else if (ch == 'C') code = 1;
        else if (ch == 'G') code = 2;
        else if (ch == 'T') code = 3;
        else refexc[(p >> 4) >> 5] |= 1u << ((p >> 4) & 31);
        ref2b[p >> 4] |= code << (2 * (p & 15));
    }
    a.ref_2bit = ref2b.data();
    a.ref_exc = refexc.data();
    a.n_regions = (uint32_t)n;
    a.max_branch_factor = cfg->max_branch_factor;
    a.enable_exact_shortcut = cfg->enable_exact_shortcut;
    a.mode = mode;
    if (mode == 1) { /* the pre-checks of avk_optimize_pairs_batch (aardvark_amd/csrc/avk_host.hip) */
        for (uint64_t r = 0; r < batch->n_regions; ++r) {
            AvkDevRegion &dr = pb.regions[r];
            if ((dr.pre_status & 0xFFFFu) == AVK_ST_INVALID_INPUT) continue;
            if (pb.zyg_flags[r] & 1) dr.pre_status = AVK_ST_BAD_ZYGOSITY;
            else if (pb.delta_t[r] != pb.delta_q[r]) dr.pre_status = AVK_PRE_SKIP_OK;
            else if (pb.zyg_flags[r] & 2) dr.pre_status = AVK_ST_BAD_ZYGOSITY;
            else dr.pre_status = 0;
        }
    }
    a.tier[0].ws_bytes = lds_bytes;
    a.tier[0].ed_cap = lds_ed_cap;
    a.tier[1].ws_bytes = lds2_bytes;
    a.tier[1].ed_cap = lds2_ed_cap;
    a.tier[2].ws_bytes = ws_bytes;
    a.tier[2].ed_cap = 0;
    a.tier[3].ws_bytes = big_ws_bytes;
    a.tier[3].ed_cap = 0;
    a.region_out = rout.data();
    a.group_metrics = out->group_metrics ? gm.data() : nullptr;
    a.var_out = vout.data();
    a.seq_bytes = want_seq ? out->seq_bytes : nullptr;
    a.seq_len = want_seq ? out->seq_len : nullptr;
    a.tally = partials.data();

    if (cfg->max_branch_factor == 0) { /* query_optimizer.rs:177 */
        for (uint64_t r = 0; r < n; ++r)
            if (!(pb.regions[r].pre_status & 0xFFFFu)) pb.regions[r].pre_status = AVK_ST_BRANCH_FACTOR;
    }

    /* solo_waves extra waves run before the others with the tier-1 slice size (the first solo_blocks workgroups of the launch) */
    AvkKernelArgs a_solo;
    auto run_pass = [&](uint32_t waves, uint64_t slice_bytes, uint64_t lds, uint32_t solo_waves = 0) {
        if (threads < 1) threads = 1;
        a.n_waves = waves;
        std::vector<uint8_t> hbm(slice_bytes ? (size_t)waves * slice_bytes : 0);
        a.hbm_ws = slice_bytes ? hbm.data() : nullptr;
        std::atomic<uint32_t> next(0);
        auto worker = [&]() {
            avk_emu::Wave w;
            w.stack_bytes = 256 * 1024;
            std::vector<char> stacks(64 * w.stack_bytes + 64);
            w.stacks = stacks.data();
            std::vector<uint8_t> ldsbuf(lds ? (solo_waves && lds2_bytes > lds ? lds2_bytes : lds) : 8);
            for (;;) {
                uint32_t wid = next.fetch_add(1);
                if (wid >= (a.esc_bytes ? 0u : waves) + solo_waves) break;
                const bool solo = wid < solo_waves;
                WaveTask t{solo ? &a_solo : &a, solo ? wid : wid - solo_waves, lds ? ldsbuf.data() : nullptr};
                avk_emu::run_wave(&w, lane_main, &t);
            }
        };
        /* in-workgroup escalation: the 4 waves of a workgroup run side by side (one OS thread each) on one LDS buffer */
        std::atomic<uint32_t> next_wg(0);
        auto wg_worker = [&]() {
            const uint32_t n_wg = (waves + 3) / 4;
            std::vector<uint8_t> wgbuf(4 * lds + 64);
            for (;;) {
                const uint32_t g = next_wg.fetch_add(1);
                if (g >= n_wg) break;
                const uint32_t alive = waves - 4 * g < 4 ? waves - 4 * g : 4;
                uint32_t *ctl = (uint32_t *)(wgbuf.data() + a.esc_bytes);
                for (int k = 0; k < AVK_WG_TAIL_BYTES / 4; ++k) ctl[k] = 0;
                for (uint32_t k = alive; k < 4; ++k) ctl[2 + k] = 0xFFFFFFFFu; /* a short last workgroup: the missing waves never park */
                ctl[6] = 4 - alive;                                            /* ... and count as gone */
                std::vector<std::thread> wt;
                for (uint32_t k = 0; k < alive; ++k)
                    wt.emplace_back([&, k]() {
                        avk_emu::Wave w;
                        w.stack_bytes = 256 * 1024;
                        std::vector<char> stacks(64 * w.stack_bytes + 64);
                        w.stacks = stacks.data();
                        WaveTask t{&a, 4 * g + k, wgbuf.data() + (size_t)k * a.tier[0].ws_bytes};
                        t.wg_lds = wgbuf.data();
                        t.wave_in_wg = k;
                        t.n_wg_waves = alive;
                        avk_emu::run_wave(&w, lane_main, &t);
                    });
                for (auto &x : wt) x.join();
            }
        };
        std::vector<std::thread> ts;
        for (int i = 0; i < threads; ++i) ts.emplace_back(worker);
        for (auto &t : ts) t.join();
        if (a.esc_bytes && lds) {
            std::vector<std::thread> tg;
            for (int i = 0; i < (threads + 3) / 4; ++i) tg.emplace_back(wg_worker);
            for (auto &t : tg) t.join();
        }
    };

    /* the same four tier launches as avk_compare_resident (aardvark_amd/csrc/avk_host.hip) */
    const bool use[4] = {lds_bytes > 0, lds2_bytes > 0, ws_bytes > 0, big_ws_bytes > 0};
    const bool launch[4] = {use[0], use[1] && (lds2_overflow_pass || !use[0]), use[2], use[3] && !use[2]};
    int last = -1;
    for (int t = 0; t < 4; ++t)
        if (launch[t]) last = t;
    if (last < 0) return AVK_E_ARG;
    const uint32_t big_slots = use[2] && use[3] ? 2u : 0u;
    std::vector<uint8_t> big_slices(big_slots ? (size_t)big_slots * big_ws_bytes : 0);
    /* work order and solo waves as in upload_internal / run_internal (aardvark_amd/csrc/avk_host.hip) */
    std::vector<uint32_t> order;
    const avk::WorkPlan plan = avk::plan_work_order(pb, avk::bulk_slice_bytes(lds_bytes), lds_ed_cap, lds2_bytes, lds2_ed_cap, mode == 1 ? 0u : solo_min_variants, 50, &order,
                                                    getenv("AVK_EMU_CLASS_C") ? (uint32_t)atoi(getenv("AVK_EMU_CLASS_C")) : 12u,
                                                    g_lane_kernel ? 0ull : 0xFFFFFFFFull /* as upload_internal does with the option lane_kernel off */);
    const avk::PodVec<AvkDevRegion> sorted = avk::regions_in_work_order(pb, order); /* the records go in work order */
    a.regions = sorted.data();
    /* the lane-per-region launches of run_internal (aardvark_amd/csrc/avk_host.hip): fast segments first, leftovers to the list the
     * first HBM pass reads */
    const int fast_list = launch[1] ? 1 : 0;
    const bool use_fast = g_lane_kernel && launch[0] && !launch[1] && launch[2] && !launch[3] && plan.n_fast_total && !cfg->enable_sequences && !cfg->enable_exact_shortcut && n;
    (void)fast_list;
    const uint32_t n_fast = use_fast ? plan.n_fast_total : 0u;
    g_lane_solved = 0;
    if (use_fast) {
        uint64_t word_base[AVK_FAST_CLASSES];
        uint32_t n_tiles[AVK_FAST_CLASSES];
        const avk::PodVec<uint32_t> fast = avk::build_fast_records(pb, order, plan, word_base, n_tiles);
        AvkKernelArgs f = a;
        f.overflow_list = lists[2].data(); /* the DEFERRED list: an LDS pass of the wave-per-region code after the bulk */
        f.overflow_count = counters + 1024 + 32;
        for (int fc = AVK_FAST_CLASSES - 1; fc >= 0; --fc) {
            if (!n_tiles[fc]) continue;
            const AvkFastClass &cl = AVK_FAST_CLASS[fc];
            avk::lane::LaneArgs la;
            la.recs = fast.data() + word_base[fc];
            la.rec_words = AVK_FAST_WORDS_OF(cl.maxv);
            la.n_tiles = n_tiles[fc];
            la.tile_counter = counters + 1220 + fc;
            la.W = cl.W;
            la.nm = 1u << cl.maxv;
            la.ed_max = cl.ed_max;
            la.qcap = cl.qcap;
            la.gen_base = plan.fast_base[fc];
            la.lanes_log2 = g_lane_width_log2[cl.maxv - 1];
            la.max_nodes = cl.maxv > 2 ? g_lane_node_cap : 250u;
            la.max_ed_c = 0;
            AvkKernelArgs f3 = f; /* run_internal: the three-call class hands back to a list of its own, solved by an HBM-tier launch right behind it */
            f3.overflow_list = lists[3].data();
            f3.overflow_count = counters + 1104;
            auto launch = [&](const avk::lane::LaneArgs &la) {
                const uint32_t rows = avk::lane::lane_rows(la.W, la.nm, la.ed_max, la.qcap);
                std::atomic<uint32_t> next(0);
                const uint32_t waves = n_waves ? n_waves : 1;
                const int nthr = threads < 1 ? 1 : threads;
                std::vector<uint64_t> sums((size_t)nthr * AVK_TALLY_STRIDE, 0);
                auto worker = [&](int tid) {
                    avk_emu::Wave w;
                    w.stack_bytes = 256 * 1024;
                    std::vector<char> stacks(64 * w.stack_bytes + 64);
                    w.stacks = stacks.data();
                    std::vector<uint32_t> lds(((size_t)rows << la.lanes_log2) + 64, 0xA5A5A5A5u), tl(288, 0);
                    uint64_t *sm = sums.data() + (size_t)tid * AVK_TALLY_STRIDE;
                    for (;;) {
                        const uint32_t wid = next.fetch_add(1);
                        if (wid >= waves) break;
                        LaneTask t;
                        t.args = cl.maxv > 2 ? &f3 : &f;
                        t.la = &la;
                        t.wave_id = wid;
                        t.lds = lds.data();
                        t.tally = tl.data();
                        avk_emu::run_wave(&w, lane_kernel_main, &t);
                        for (int l = 0; l < 64; ++l) {
                            sm[AVK_TALLY_SOLVED] += t.n_ok[l];
                            sm[AVK_TALLY_ERRORS] += t.n_err[l];
                            sm[AVK_TALLY_LANE_SOLVED] += t.n_ok[l] + t.n_err[l];
                        }
                    }
                    for (int i = 0; i < AVK_N_GROUPS * AVK_N_FIELDS; ++i) sm[i] += tl[i];
                };
                std::vector<std::thread> ts;
                for (int i = 0; i < nthr; ++i) ts.emplace_back(worker, i);
                for (auto &t : ts) t.join();
                for (int i = 0; i < nthr; ++i)
                    for (int k = 0; k < AVK_TALLY_STRIDE; ++k) partials[k] += sums[(size_t)i * AVK_TALLY_STRIDE + k];
            };
            /* run_internal: the head of the class (regions with estimated edits) in narrow tiles of its own, ahead of the rest */
            const uint32_t head_tiles = g_lane_head_width ? (plan.n_fast_heavy[fc] + 63u) / 64u : 0u;
            if (head_tiles > 0 && head_tiles < la.n_tiles && g_lane_head_width < (1u << la.lanes_log2)) {
                avk::lane::LaneArgs hd = la;
                hd.n_tiles = head_tiles;
                hd.tile_counter = counters + 1230 + fc;
                hd.lanes_log2 = g_lane_head_width <= 4 ? 2u : (g_lane_head_width <= 8 ? 3u : (g_lane_head_width <= 16 ? 4u : 5u));
                launch(hd);
                la.recs += (size_t)head_tiles * la.rec_words * 64u;
                la.n_tiles -= head_tiles;
                la.gen_base += head_tiles * 64u;
            }
            launch(la);
            if (cl.maxv > 2 && counters[1104]) { /* run_internal: the HBM-tier launch behind the three-call class */
                AvkKernelArgs keep = a;
                a.pass_tier = 2;
                a.work_list = lists[3].data();
                a.n_work_dev = counters + 1104;
                a.work_base = 0;
                a.n_work = 0;
                a.work_counter = counters + 1120;
                a.static_pct = 0;
                a.n_shards = 1;
                a.claim = 1;
                a.esc_bytes = 0;
                a.esc_enabled = 0;
                a.high_priority = 0;
                a.extra_counter = nullptr;
                a.extra_n = 0;
                a.overflow_list = nullptr;
                a.overflow_count = nullptr;
                a.big_ws = big_slots ? big_slices.data() : nullptr;
                a.big_busy = counters + 1088;
                a.big_slots = big_slots;
                run_pass(n_waves ? n_waves : 1, ws_bytes, 0);
                a = keep;
            }
        }
    }
    const uint32_t *list = nullptr, *count = nullptr;
    int nlist = 0;
    uint32_t hbm_shared = 0; /* length of the class C list the HBM launches share */
    for (int t = 0; t < 4 && n; ++t) {
        if (!launch[t]) continue;
        a.pass_tier = (uint32_t)t;
        a.big_slots = 0;
        a.work_list = list;
        a.work_base = 0;
        a.n_work_dev = count;
        a.n_work = (uint32_t)n - (t == 0 ? n_fast : 0u);
        a.high_priority = 0;
        a.esc_bytes = 0;
        a.esc_enabled = 0;
        a.static_pct = AVK_STATIC_PCT;
        a.n_shards = 8;
        a.claim = AVK_CLAIM;
        a.work_counter = counters + 256 * t;
        if (t != last) {
            a.overflow_list = lists[nlist].data();
            a.overflow_count = counters + 1024 + 16 * nlist;
        } else {
            a.overflow_list = nullptr;
            a.overflow_count = nullptr;
        }
        const uint32_t todo = count ? *count : (uint32_t)n;
        if (todo || (t == 2 && hbm_shared)) {
            if (t == 0) {
                /* the three concurrent launches of run_internal, one after the other: HBM solo (class C), LDS solo (class B), bulk */
                const bool solo_ok = solo_min_variants != 0;
                const uint32_t n_c = solo_ok && launch[2] ? plan.n_hbm : 0u;
                const uint32_t n_front = plan.n_hbm + plan.n_hard - n_c;
                const int solo_list = launch[1] ? 1 : 0;
                const bool later = last > solo_list;
                /* AVK_EMU_SKIP_HBM_SOLO=1 (tests): the solo launch starts nothing, so the whole class C list is left to the shared
                 * ticket counter of the main HBM pass — the state of a GPU run whose bulk finished before the solo launch got going */
                const bool skip_hbm_solo = getenv("AVK_EMU_SKIP_HBM_SOLO") != nullptr;
                hbm_shared = n_c;
                if (n_c && !skip_hbm_solo) {
                    AvkKernelArgs keep = a;
                    a.pass_tier = 2;
                    a.work_base = 0;
                    a.n_work = n_c;
                    a.work_counter = counters + 1076;
                    a.static_pct = 0;
                    a.n_shards = 1;
                    a.claim = 1;
                    a.high_priority = 1;
                    a.big_ws = big_slots ? big_slices.data() : nullptr;
                    a.big_busy = counters + 1088;
                    a.big_slots = big_slots;
                    a.overflow_list = nullptr;
                    a.overflow_count = nullptr;
                    if (!big_slots && launch[3]) {
                        a.overflow_list = lists[launch[1] ? 2 : 1].data();
                        a.overflow_count = counters + 1024 + 16 * (launch[1] ? 2 : 1);
                    }
                    run_pass(n_waves / 2 ? n_waves / 2 : 1, ws_bytes, 0);
                    a = keep;
                }
                uint32_t solo = 0;
                if (solo_ok && use[1] && n_front && lds2_bytes >= lds_bytes) {
                    solo = n_waves / 4 ? n_waves / 4 : 1;
                    if (solo > n_front) solo = n_front;
                    a_solo = a;
                    a_solo.pass_tier = 1;
                    a_solo.work_base = n_c;
                    a_solo.n_work = solo;
                    a_solo.work_counter = counters + 1072;
                    a_solo.static_pct = 0;
                    a_solo.n_shards = 1;
                    a_solo.claim = 1;
                    a_solo.n_waves = solo;
                    a_solo.high_priority = 1;
                    a_solo.overflow_list = later ? lists[solo_list].data() : nullptr;
                    a_solo.overflow_count = later ? counters + 1024 + 16 * solo_list : nullptr;
                }
                a.work_base = n_c + solo;
                a.n_work = (uint32_t)n - n_fast - n_c - solo;
                if (lds_bytes >= 1024) {
                    a.tier[0].ws_bytes = avk::bulk_slice_bytes(lds_bytes);
                    a.esc_bytes = (uint32_t)(4 * a.tier[0].ws_bytes);
                    a.esc_enabled = lds_escalation ? 1u : 0u;
                }
                run_pass(n_waves ? n_waves : 1, 0, lds_bytes, solo);
                if (use_fast && counters[1024 + 32]) { /* what the lanes handed over (run_internal: the deferred LDS launch) */
                    AvkKernelArgs keep = a;
                    a.work_list = lists[2].data();
                    a.n_work_dev = counters + 1024 + 32;
                    a.work_base = 0;
                    a.n_work = 0;
                    a.work_counter = counters + 768;
                    a.overflow_list = lists[1].data(); /* its own overflow list: one more HBM pass at the very end */
                    a.overflow_count = counters + 1024 + 16;
                    run_pass(n_waves ? n_waves : 1, 0, lds_bytes, 0);
                    a = keep;
                }
                a.tier[0].ws_bytes = lds_bytes;
                a.esc_bytes = 0;
                a.esc_enabled = 0;
            }
            else if (t == 1) run_pass(n_waves ? n_waves : 1, 0, lds2_bytes);
            else if (t == 2) {
                a.big_ws = big_slots ? big_slices.data() : nullptr;
                a.big_busy = counters + 1088;
                a.big_slots = big_slots;
                if (hbm_shared) { /* run_internal: the class C list is shared with the HBM solo launch through its ticket counter */
                    a.extra_counter = counters + 1076;
                    a.extra_base = 0;
                    a.extra_n = hbm_shared;
                }
                run_pass(n_waves ? n_waves : 1, ws_bytes, 0);
                a.extra_counter = nullptr;
                a.extra_n = 0;
            }
            else run_pass(todo < 4 ? todo : 4, big_ws_bytes, 0);
        }
        if (t != last) {
            list = lists[nlist].data();
            count = counters + 1024 + 16 * nlist;
            nlist += 1;
        }
    }

    if (use_fast && counters[1024 + 16]) { /* run_internal: what the handed-back regions' LDS pass could not hold */
        a.pass_tier = 2;
        a.work_list = lists[1].data();
        a.n_work_dev = counters + 1024 + 16;
        a.work_base = 0;
        a.n_work = 0;
        a.work_counter = counters + 256;
        a.static_pct = 0;
        a.n_shards = 1;
        a.claim = 1;
        a.esc_bytes = 0;
        a.esc_enabled = 0;
        a.high_priority = 0;
        a.extra_counter = nullptr;
        a.extra_n = 0;
        a.overflow_list = nullptr;
        a.overflow_count = nullptr;
        a.big_ws = big_slots ? big_slices.data() : nullptr;
        a.big_busy = counters + 1088;
        a.big_slots = big_slots;
        run_pass(n_waves ? n_waves : 1, ws_bytes, 0);
    }
    for (int c = 0; c < AVK_TALLY_COPIES; ++c) /* avk_tally_reduce */
        for (int i = 0; i < AVK_TALLY_STRIDE; ++i) tally[i] += partials[(size_t)c * AVK_TALLY_STRIDE + i];
    /* copy back in caller order */
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t *w = rout.data() + 4 * r;
        out->status[r] = (int32_t)w[0];
        if (out->ed_h1) out->ed_h1[r] = w[1];
        if (out->ed_h2) out->ed_h2[r] = w[2];
        if (out->n_optima) out->n_optima[r] = w[3] & 0xFFFFu;
        if (out->type_present) out->type_present[r] = (uint16_t)(w[3] >> 16);
    }
    if (out->group_metrics) memcpy(out->group_metrics, gm.data(), gm.size() * sizeof(uint32_t));
    for (uint64_t v = 0; v < nv; ++v) {
        const uint64_t hv = pb.dev2host[v];
        const uint32_t w = vout[v];
        if (out->var_expected) out->var_expected[hv] = (uint8_t)(w & 0xFF);
        if (out->var_observed) out->var_observed[hv] = (uint8_t)((w >> 8) & 0xFF);
        if (out->var_class) out->var_class[hv] = (uint8_t)((w >> 16) & 0xFF);
        if (out->var_zyg) out->var_zyg[hv] = (uint8_t)(w >> 24);
    }
    if (out->tally) memcpy(out->tally, tally.data(), AVK_TALLY_LEN * sizeof(uint64_t));
    if (tier_counts) memcpy(tier_counts, tally.data() + AVK_TALLY_LEN, 5 * sizeof(uint64_t));
    g_lane_solved = tally[AVK_TALLY_LANE_SOLVED];
    return 0;
}

/* avk_dwfa_script_batch on emulated wavefronts / lanes (same device functions, avk_dwfa_script.inl) */
struct DwfaTask {
    AvkDwfaArgs a;
    uint32_t first; /* engine 0: the script; engine 1: the first script of the wave's 64 */
    int engine;
    uint32_t *lds;
};
static void dwfa_task_main(void *p, int lane) {
    DwfaTask *t = (DwfaTask *)p;
    if (t->engine == 0) avk::dwfa_script_wave(t->a, t->first);
    else if (t->first + (uint32_t)lane < t->a.n_scripts) avk::lane::dwfa_script_lane(t->a, t->first + (uint32_t)lane, t->lds);
}
int emu_dwfa_script_batch(int engine, uint32_t n_scripts, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *base_off, const uint64_t *other_off,
                          const uint64_t *step_off, const uint8_t *step_op, const uint32_t *step_blen, const uint32_t *step_olen, uint32_t *step_ed,
                          int32_t *step_status, uint32_t wf_cap, uint32_t *final_wf, uint32_t *final_wf_len) {
    (void)n_bytes;
    std::vector<uint32_t> ws((size_t)n_scripts * wf_cap + 64), lds((size_t)avk::lane::dwfa_lane_rows() * 64 + 64, 0x5A5A5A5Au);
    DwfaTask t;
    memset(&t.a, 0, sizeof(t.a));
    t.a.bytes = bytes, t.a.base_off = base_off, t.a.other_off = other_off, t.a.step_off = step_off, t.a.step_op = step_op, t.a.step_blen = step_blen,
    t.a.step_olen = step_olen, t.a.step_ed = step_ed, t.a.step_status = step_status, t.a.final_wf = final_wf, t.a.final_wf_len = final_wf_len, t.a.ws = ws.data(),
    t.a.wf_cap = wf_cap, t.a.n_scripts = n_scripts;
    t.engine = engine;
    t.lds = lds.data();
    avk_emu::Wave w;
    w.stack_bytes = 256 * 1024;
    std::vector<char> stacks(64 * w.stack_bytes + 64);
    w.stacks = stacks.data();
    for (uint32_t s = 0; s < n_scripts; s += engine == 0 ? 1u : 64u) {
        t.first = s;
        avk_emu::run_wave(&w, dwfa_task_main, &t);
    }
    return 0;
}

void emu_set_lane_kernel(int on) { g_lane_kernel = on; }
#ifdef AVK_LANE_STATS
void emu_lane_work(uint32_t *per_region) { avk::lane::g_lane_work = per_region; }
void emu_lane_stats(uint64_t *out, int reset) {
    for (int i = 0; i < 32; ++i) {
        out[i] = avk::lane::g_lane_stats[i];
        if (reset) avk::lane::g_lane_stats[i] = 0;
    }
}
#endif
uint64_t emu_last_lane_solved(void) { return g_lane_solved; }
void emu_set_lane_node_cap(int cap) { g_lane_node_cap = (uint32_t)cap; }
void emu_set_lane_head_width(int w) { g_lane_head_width = (uint32_t)w; }
void emu_set_lane_width(int one, int two) {
    g_lane_width_log2[0] = one <= 16 ? 4 : (one <= 32 ? 5 : 6);
    g_lane_width_log2[1] = two <= 16 ? 4 : (two <= 32 ? 5 : 6);
}

int emu_compare_batch(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
                      const avk_compare_config *cfg, avk_result_batch *out, uint64_t lds_bytes, uint32_t lds_ed_cap, uint64_t lds2_bytes,
                      uint32_t lds2_ed_cap, uint64_t ws_bytes, uint64_t big_ws_bytes, uint32_t n_waves, int threads, uint64_t *tier_counts,
                      uint32_t solo_min_variants, uint32_t lds2_overflow_pass, uint32_t lds_escalation) {
    return emu_run(0, batch, refs, ref_lens, n_contigs, cfg, out, lds_bytes, lds_ed_cap, lds2_bytes, lds2_ed_cap, ws_bytes, big_ws_bytes, n_waves,
                   threads, tier_counts, solo_min_variants, lds2_overflow_pass, lds_escalation);
}

/* avk_optimize_pairs_batch on emulated wavefronts (default tier sizes) */
int emu_optimize_pairs_batch(const avk_region_batch *batch, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs,
                             uint32_t max_branch_factor, int32_t *status, uint8_t *is_exact_match, int threads) {
    avk_compare_config cfg;
    cfg.max_branch_factor = max_branch_factor;
    cfg.enable_sequences = 0;
    cfg.enable_exact_shortcut = 0;
    std::vector<uint32_t> ed1(batch->n_regions + 1);
    avk_result_batch out;
    memset(&out, 0, sizeof(out));
    out.status = status;
    out.ed_h1 = ed1.data();
    int rc = emu_run(1, batch, refs, ref_lens, n_contigs, &cfg, &out, 10 * 1024, 48, 40 * 1024, 48, 1 << 20, 64ull << 20, 8, threads, nullptr, 0, 0, 1);
    if (rc) return rc;
    for (uint64_t r = 0; r < batch->n_regions; ++r) is_exact_match[r] = status[r] == 0 && ed1[r] ? 1 : 0;
    return 0;
}

} /* extern "C" */
