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
    /* quad rendezvous (qd_bcast / qd_xor / qd_sync): generation, values and call sites per lane; the four lanes of a quad must agree */
    uint32_t qgen[64];
    uint64_t qslots[2][64];
    uint32_t qsites[2][64];
    uint64_t progress; /* deposits and finished fibers so far: a full round of the fibers without any is a deadlock */
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
/* the fibers take turns in lane order; a fiber runs until its next rendezvous (of the wave or of its quad) and waits there, passing the turn on,
 * until the others it meets there have arrived */
static int next_runnable(const Wave *w, int me) {
    for (int i = 1; i <= 64; ++i) {
        const int t = (me + i) & 63;
        if (!w->done[t] && t != me) return t;
    }
    return -1;
}
static void wait_turn(Wave *w, int me, const char *what, uint32_t site) {
    const uint64_t p = w->progress;
    const int nx = next_runnable(w, me);
    if (nx >= 0) switch_to(w, me, nx);
    if (w->progress == p) { /* every other fiber had its turn and none arrived anywhere */
        fprintf(stderr, "avk_emu: DEADLOCK: lane %d waits at a %s rendezvous (site %u) that the others never reach\n", me, what, site & 0x7FFFFFFFu);
        abort();
    }
}

const uint64_t *gather(uint64_t v, uint32_t site) {
    Wave *w = t_wave;
    const int me = w->cur;
    const uint32_t g = w->gen[me]++;
    w->slots[g & 1][me] = v;
    w->sites[g & 1][me] = site;
    w->progress += 1;
    for (;;) {
        bool all = true;
        for (int i = 0; i < 64; ++i)
            if (w->gen[i] < g + 1) {
                all = false;
                if (w->done[i]) {
                    fprintf(stderr, "avk_emu: DIVERGENT wave primitive: lane %d at site %u, lane %d has finished (gen %u/%u)\n", me, site & 0x7FFFFFFFu, i, g, w->gen[i]);
                    abort();
                }
            }
        if (all) break;
        wait_turn(w, me, "wave", site);
    }
    /* every lane has deposited generation g */
    const uint32_t *s = w->sites[g & 1];
    for (int i = 0; i < 64; ++i) {
        if (s[i] != site) {
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

const uint64_t *quad_gather(uint64_t v, uint32_t site) {
    Wave *w = t_wave;
    const int me = w->cur, q0 = me & ~3;
    const uint32_t g = w->qgen[me]++;
    w->qslots[g & 1][me] = v;
    w->qsites[g & 1][me] = site;
    w->progress += 1;
    for (;;) {
        bool all = true;
        for (int i = q0; i < q0 + 4; ++i)
            if (w->qgen[i] < g + 1) {
                all = false;
                if (w->done[i]) {
                    fprintf(stderr, "avk_emu: DIVERGENT quad primitive: lane %d at site %u, lane %d has finished\n", me, site, i);
                    abort();
                }
            }
        if (all) break;
        wait_turn(w, me, "quad", site);
    }
    for (int i = q0; i < q0 + 4; ++i)
        if (w->qsites[g & 1][i] != site) {
            fprintf(stderr, "avk_emu: DIVERGENT quad primitive: lane %d at site %u, lane %d at site %u\n", me, site, i, w->qsites[g & 1][i]);
            abort();
        }
    return w->qslots[g & 1] + q0;
}

static void fiber_entry() {
    Wave *w = t_wave;
    const int me = w->cur;
    w->fn(w->arg, me);
    w->done[me] = true;
    w->progress += 1;
    const int nx = next_runnable(w, me);
    void *dummy;
    if (nx < 0) { /* the last fiber: all lanes must have passed the same number of wave rendezvous */
        for (int i = 0; i < 64; ++i)
            if (w->gen[i] != w->gen[me]) {
                fprintf(stderr, "avk_emu: lane %d finished after %u rendezvous but lane %d after %u\n", me, w->gen[me], i, w->gen[i]);
                abort();
            }
        w->cur = -1;
        avk_emu_switch(&dummy, w->main_sp);
    } else {
        w->cur = nx;
        avk_emu_switch(&dummy, w->sp[nx]);
    }
    abort(); /* unreachable */
}

/* runs fn(arg, lane) on 64 lanes as one wavefront */
static void run_wave(Wave *w, void (*fn)(void *, int), void *arg) {
    w->fn = fn;
    w->arg = arg;
    w->progress = 0;
    for (int i = 0; i < 64; ++i) {
        w->gen[i] = 0;
        w->qgen[i] = 0;
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
namespace avk { namespace lane { uint64_t g_lane_stats[32]; int g_lane_phase; uint32_t *g_lane_work; uint32_t *g_lane_comp; } }
#endif
#include "../../aardvark_amd/csrc/avk_lane.inl"
#include "../../aardvark_amd/csrc/avk_quad.inl"
#ifdef AVK_WIDE_STATS
namespace avk { namespace wide { uint64_t g_wide_defer[64]; uint32_t g_wide_defer_region[64]; } }
#endif
#include "../../aardvark_amd/csrc/avk_wide.inl"
#include "../../aardvark_amd/csrc/avk_dwfa_script.inl"
#include "../../aardvark_amd/csrc/avk_devpack.inl"

namespace {

int g_lane_kernel = 1; /* emu_set_lane_kernel: small regions through the lane-per-region code (avk_lane.inl), as run_internal does */
uint64_t g_lane_solved = 0;
uint32_t g_lane_width_log2[3] = {6, 6, 4};
uint32_t g_lane_node_cap = 32;
int g_lane_pool = -1; /* context option lane_pool */
int g_pair_classes = 1; /* context option pair_classes */
uint32_t g_lane_head_width = 16; /* run_internal's option lane_head_width */ /* emu_set_lane_width: records a wave takes at a time in the one-call / two-call classes */

struct LaneTask {
    const AvkKernelArgs *args;
    const avk::lane::LaneArgs *la;
    uint32_t wave_id;
    uint32_t *lds, *tally;
    uint32_t n_ok[64], n_err[64];
};
struct PairTask {
    const AvkKernelArgs *args;
    const avk::pairs::PairArgs *pa;
    uint64_t *part;
};
void pair_kernel_main(void *p, int) {
    PairTask *t = (PairTask *)p;
    avk::pairs::pair_worker(*t->args, *t->pa, t->part);
}
void lane_kernel_main(void *p, int lane) {
    LaneTask *t = (LaneTask *)p;
    uint32_t ok = 0, err = 0;
    avk::lane::lane_worker(*t->args, *t->la, t->wave_id, t->lds, t->tally, ok, err);
    t->n_ok[lane] = ok;
    t->n_err[lane] = err;
}
uint64_t g_quad_solved = 0; /* emu_last_quad_solved: regions the quad launches of the last call finished */
int g_lane_quad = 1; /* context option lane_quad: launches of at most 16 records per wave (the heads, the three-call class) run four lanes per region (avk_quad.inl) */
void quad_kernel_main(void *p, int lane) {
    LaneTask *t = (LaneTask *)p;
    uint32_t ok = 0, err = 0;
    avk::quad::quad_worker(*t->args, *t->la, t->wave_id, t->lds, t->tally, ok, err);
    t->n_ok[lane] = ok;
    t->n_err[lane] = err;
}

int g_wide_kernel = 1;                 /* emu_set_wide_kernel: context option wide_kernel (avk_wide.inl ahead of the HBM-tier launches of class C and of the three-call class's hand-backs) */
uint32_t g_wide_lds_bytes = 16 * 1024; /* context option wide_lds_bytes */
uint64_t g_wide_solved = 0;
struct WideTask {
    const AvkKernelArgs *args;
    avk::wide::WideArgs wa;
    uint32_t wave_id;
    uint32_t *lds;
};
void wide_kernel_main(void *p, int) {
    WideTask *t = (WideTask *)p;
    avk::wide::wide_worker<true>(*t->args, t->wa, t->wave_id, t->lds); /* (the lazy instantiation: it differs from the other only when lazy_dp is set) */
}

struct WaveTask {
    const AvkKernelArgs *args;
    uint32_t wave_id;
    uint8_t *lds;
    uint8_t *wg_lds = nullptr; /* the workgroup's whole LDS when the in-workgroup escalation is on */
    uint32_t wave_in_wg = 0, n_wg_waves = 0;
};

int g_team = 0; /* emu_set_team: the HBM-tier passes run their regions as avk_region_kernel_team's owner wave does — the search posts its independent pieces as jobs
                   (children made out of place, the metrics' alignments one job each).  The emulator runs one wave at a time, so the owner takes every job itself:
                   the decomposition is what is checked here, the hand-over between waves on the GPU (tests/test_gpu_team.py). */
void lane_main(void *p, int /*lane*/) {
    WaveTask *t = (WaveTask *)p;
    if (t->lds) avk::region_worker<true, true>(*t->args, t->wave_id, t->lds); /* (the lazy instantiations: they differ from the others only when lazy_dp is set) */
    else if (g_team) {
        static thread_local avk::TeamBox box; /* (one wave = one thread of the emulator's pool at a time) */
        avk::region_worker<false, true, true>(*t->args, t->wave_id, nullptr, &box);
    } else avk::region_worker<false, true>(*t->args, t->wave_id, nullptr);
}

/* ---- the device-side packer (aardvark_amd/csrc/avk_devpack.inl) run the way upload_device_packed of avk_devpack_host.inl queues it: the same
 * device functions, the workgroup-level plumbing (histogram, scans, scatter) as plain loops */
uint64_t g_last_pair_regions = 0; /* emu_last_pair_regions: regions of the last emulated call that went through the lookup of avk_pairs.inl */
uint32_t g_hbm_ed_cap = 1024; /* emu_set_hbm_ed_cap: context option hbm_ed_cap */
int g_lane_pairs = 1; /* emu_set_lane_pairs: context option lane_pairs (regions with the same SNV on both sides are looked up, avk_pairs.inl) */
uint64_t g_class_c_below = 0; /* emu_set_class_c_below: context option class_c_below (0 = no such rule, the emulator's default) */
uint32_t g_stripe_w = 0; /* emu_set_stripe: claim width the heads of the lane classes are dealt out over (context option lane_stripe; 0 = sorted order) */
int g_last_packed_source = 0; /* emu_last_packed_source: the last packing run read its batch from the packed form */
int g_dp_packed_source = 0; /* emu_set_packed_source: the device packer reads batches that fit the packed form from packed arrays (DpIn::pk_*), as the library does for avk_packed_batch */
int g_device_pack = 0; /* emu_set_device_pack: emu_run packs its batch with the device functions instead of avk_pack.h */
namespace dpk = avk::dp;
struct DpResult {
    std::vector<AvkDevRegion> regions; /* work order */
    std::vector<uint32_t> blob, fast, order, v_off, blob_off8, bp_off;
    std::vector<uint64_t> seq_off;
    std::vector<dpk::DpVarInfo> vinfo;
    std::vector<dpk::DpRegionInfo> rinfo;
    std::vector<dpk::DpSlot> slots;
    std::vector<uint32_t> pending, big_list;
    std::vector<uint8_t> owned;
    dpk::DpState st;
    dpk::DpArgs args; /* for the writers that run later (records of the regions the lanes hand back) */
    bool lazy = false;
    bool packed_source = false; /* the packer read the batch from its packed form (g_dp_packed_source and the batch fits it) */
    std::vector<uint32_t> pk_start;
    std::vector<uint16_t> pk_len, pk_contig, pk_rel;
    std::vector<uint8_t> pk_tc, pk_qc, pk_tz, pk_a0, pk_a1;
    std::vector<uint64_t> pk_voff, pk_aoff;
    std::string err;
};
struct DpWaveTask {
    const dpk::DpArgs *a;
    uint32_t item;
};
void dp_wave_main(void *p, int /*lane*/) {
    DpWaveTask *t = (DpWaveTask *)p;
    dpk::dp_region_record_wave(*t->a, t->item);
}
int dp_run(const avk_region_batch *b, const std::vector<uint64_t> &base, const std::vector<uint64_t> &lens, const dpk::DpOpts &opt, bool pairs_mode, DpResult *R, bool lazy = false) {
    const uint64_t n = b->n_regions, nv = b->n_variants;
    dpk::DpArgs a;
    memset(&a, 0, sizeof(a));
    a.in.contig_idx = b->contig_idx, a.in.start = b->start, a.in.end = b->end, a.in.t_off = b->t_off, a.in.q_off = b->q_off, a.in.t_cnt = b->t_cnt, a.in.q_cnt = b->q_cnt,
    a.in.var_pos = b->var_pos, a.in.var_type = b->var_type, a.in.var_zyg = b->var_zyg, a.in.var_raw = b->var_raw_space, a.in.a0_off = b->a0_off, a.in.a1_off = b->a1_off,
    a.in.a0_len = b->a0_len, a.in.a1_len = b->a1_len, a.in.alleles = b->allele_bytes, a.in.n_regions = n, a.in.n_variants = nv, a.in.alleles_len = b->allele_bytes_len,
    a.in.contig_base = base.data(), a.in.contig_len = lens.data(), a.in.n_contigs = (uint32_t)lens.size(), a.in.pairs_mode = pairs_mode ? 1u : 0u;
    a.in.v_lo = 0, a.in.v_hi = nv;
    R->owned.assign(nv + 16, 0); /* upload_device_packed: forms with explicit offsets count the ownership of their calls (and re-derive a shared call's position per region) */
    if (!pairs_mode) a.in.owned = R->owned.data();
    if (g_dp_packed_source && !pairs_mode) {
        /* upload_device_packed on an avk_packed_batch (round 6): the packer reads the packed arrays as they came (DpIn::pk_*), there are no wide arrays.  The batch is
         * put into that form here when it fits it (calls and alleles back to back in region order, narrow fields wide enough); otherwise the wide path above stands. */
        bool fits = n > 0;
        uint64_t run_v = 0, run_a = 0;
        for (uint64_t r = 0; r < n && fits; ++r) {
            fits = b->t_off[r] == run_v && b->q_off[r] == run_v + b->t_cnt[r] && b->t_cnt[r] < 256 && b->q_cnt[r] < 256 && b->end[r] >= b->start[r] && b->end[r] - b->start[r] < 65536 &&
                   b->start[r] < (1ull << 32) && (!b->contig_idx || b->contig_idx[r] < 65536);
            for (uint64_t v = run_v; v < run_v + b->t_cnt[r] + b->q_cnt[r] && fits; ++v)
                fits = v < nv && b->var_pos[v] >= b->start[r] && b->var_pos[v] - b->start[r] < 65536 && b->a0_len[v] < 256 && b->a1_len[v] < 256 && b->var_type[v] < 16 && b->var_zyg[v] < 16;
            run_v += (uint64_t)b->t_cnt[r] + b->q_cnt[r];
        }
        fits = fits && run_v == nv;
        for (uint64_t v = 0; v < nv && fits; ++v) {
            fits = b->a0_off[v] == run_a && b->a1_off[v] == run_a + b->a0_len[v];
            run_a += (uint64_t)b->a0_len[v] + b->a1_len[v];
        }
        fits = fits && run_a == b->allele_bytes_len;
        R->packed_source = fits;
        g_last_packed_source = fits ? 1 : 0;
        if (fits) {
            R->pk_start.assign(n + 1, 0), R->pk_len.assign(n + 1, 0), R->pk_contig.assign(n + 1, 0), R->pk_rel.assign(nv + 1, 0), R->pk_tc.assign(n + 1, 0), R->pk_qc.assign(n + 1, 0);
            R->pk_tz.assign(nv + 1, 0), R->pk_a0.assign(nv + 1, 0), R->pk_a1.assign(nv + 1, 0), R->pk_voff.assign(n + 1, 0), R->pk_aoff.assign(nv + 1, 0);
            for (uint64_t r = 0; r < n; ++r) {
                R->pk_start[r] = (uint32_t)b->start[r], R->pk_len[r] = (uint16_t)(b->end[r] - b->start[r]), R->pk_contig[r] = (uint16_t)(b->contig_idx ? b->contig_idx[r] : 0);
                R->pk_tc[r] = (uint8_t)b->t_cnt[r], R->pk_qc[r] = (uint8_t)b->q_cnt[r], R->pk_voff[r] = b->t_off[r];
                for (uint64_t v = b->t_off[r]; v < b->t_off[r] + b->t_cnt[r] + b->q_cnt[r]; ++v) R->pk_rel[v] = (uint16_t)(b->var_pos[v] - b->start[r]);
            }
            for (uint64_t v = 0; v < nv; ++v)
                R->pk_tz[v] = (uint8_t)(b->var_type[v] | (b->var_zyg[v] << 4)), R->pk_a0[v] = (uint8_t)b->a0_len[v], R->pk_a1[v] = (uint8_t)b->a1_len[v], R->pk_aoff[v] = b->a0_off[v];
            a.in.contig_idx = nullptr, a.in.start = a.in.end = a.in.t_off = a.in.q_off = nullptr, a.in.t_cnt = a.in.q_cnt = nullptr, a.in.var_pos = nullptr, a.in.var_type = a.in.var_zyg = nullptr;
            a.in.a0_off = a.in.a1_off = nullptr, a.in.a0_len = a.in.a1_len = nullptr;
            a.in.owned = nullptr; /* (the packed form owns its calls by construction) */
            a.in.pk_start = R->pk_start.data(), a.in.pk_len = R->pk_len.data(), a.in.pk_contig = b->contig_idx ? R->pk_contig.data() : nullptr, a.in.pk_rel = R->pk_rel.data();
            a.in.pk_tc = R->pk_tc.data(), a.in.pk_qc = R->pk_qc.data(), a.in.pk_tz = R->pk_tz.data(), a.in.pk_a0 = R->pk_a0.data(), a.in.pk_a1 = R->pk_a1.data();
            a.in.pk_voff = R->pk_voff.data(), a.in.pk_aoff = R->pk_aoff.data();
        }
    }
    a.opt = opt;
    R->vinfo.assign(nv + 1, dpk::DpVarInfo());
    R->rinfo.assign(n + 1, dpk::DpRegionInfo());
    R->slots.assign(nv + 1, dpk::DpSlot());
    std::vector<uint32_t> &pending = R->pending, &big_list = R->big_list;
    pending.assign(nv + 1, 0), big_list.assign(n + 1, 0);
    R->v_off.assign(n + 1, 0), R->blob_off8.assign(n + 1, 0), R->seq_off.assign(n + 1, 0), R->order.assign(n + 1, 0), R->bp_off.assign(n + 2, 0);
    memset(&R->st, 0, sizeof(R->st));
    a.vinfo = R->vinfo.data(), a.rinfo = R->rinfo.data(), a.st = &R->st, a.pending = pending.data(), a.v_off = R->v_off.data(), a.blob_off8 = R->blob_off8.data(),
    a.seq_off = R->seq_off.data(), a.order = R->order.data(), a.big_list = big_list.data(), a.bp_off = R->bp_off.data(), a.slots = R->slots.data();
    for (uint64_t v = 0; v < nv; ++v) dpk::dp_variant(a, v);
    auto region_passes = [&] {
        uint64_t run_v = 0, run_b = 0, run_s = 0, run_g = 0;
        for (uint64_t r = 0; r < n; ++r) {
            uint32_t nc, bw, fc, nb, ng;
            uint64_t sq;
            dpk::dp_region(a, r, nc, bw, sq, fc, nb, ng);
            R->bp_off[r] = (uint32_t)run_g;
            run_g += ng;
            R->bp_off[r + 1] = (uint32_t)run_g;
            if (fc) R->st.have[fc - 1] += 1;
            if (nb != 0xFFu) R->st.need_hist[nb] += 1;
            R->v_off[r] = (uint32_t)run_v, R->blob_off8[r] = (uint32_t)(run_b / 2), R->seq_off[r] = run_s;
            run_v += nc, run_b += bw, run_s += sq;
        }
        R->st.total_v = run_v, R->st.total_blob_words = run_b, R->st.total_seq = run_s, R->st.total_groups = run_g;
        dpk::dp_lane_switch(a);
        for (uint64_t r = 0; r < n; ++r) {
            const uint32_t bk = dpk::dp_bucket_of(a, r);
            R->rinfo[r].bucket = bk;
            R->st.hist[bk] += 1;
        }
        dpk::dp_bucket_bases(a);
        for (uint64_t r = 0; r < n; ++r) R->order[dpk::dp_order_slot(a, R->rinfo[r].bucket, R->st.cursor[R->rinfo[r].bucket]++)] = (uint32_t)r;
    };
    region_passes();
    if (R->st.n_pending) { /* upload_device_packed: the host's edit distance for the calls dp_variant left, then the region passes again */
        for (uint32_t k = 0; k < R->st.n_pending; ++k) {
            const uint64_t v = pending[k];
            R->vinfo[v].alt_ed = (uint32_t)avk::host_edit_distance(b->allele_bytes + b->a0_off[v], b->a0_len[v], b->allele_bytes + b->a1_off[v], b->a1_len[v]);
            R->vinfo[v].flags &= ~(uint32_t)dpk::DP_VF_PENDING;
        }
        memset(&R->st, 0, sizeof(R->st));
        region_passes();
    }
    if (R->st.err & dpk::DP_ERR_RANGE) R->err = "variant range of a region exceeds n_variants";
    else if (R->st.err & dpk::DP_ERR_ALLELE) R->err = "allele range exceeds allele_bytes_len";
    else if (R->st.err & dpk::DP_ERR_BLOB) R->err = "region blob exceeds 2 GiB; split the region's alleles";
    else if (R->st.total_v > 0x7FFFFFFFull) R->err = "more than 2^31 variant records; split the batch";
    if (!R->err.empty()) return AVK_E_ARG;
    R->regions.resize(n + 1);
    R->blob.assign((size_t)R->st.total_blob_words + 4, 0xA5A5A5A5u); /* device memory is not zeroed: the writers must write every byte that is read */
    R->fast.assign((size_t)R->st.fast_words + 64, 0xA5A5A5A5u);
    a.regions = R->regions.data(), a.blob = R->blob.data(), a.fast = R->fast.data();
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc)
        for (uint32_t t = 0; t < R->st.fast_tiles[fc]; ++t)
            for (uint32_t lane = 0; lane < 64; ++lane) dpk::dp_fast_record(a, (uint32_t)fc, t, lane);
    /* upload_device_packed: records and blobs of the regions outside the lane classes now, of the others when a launch asks for them */
    for (uint64_t k = 0; k < (lazy ? n - R->st.n_fast_total : n); ++k) dpk::dp_region_record(a, k);
    R->args = a;
    R->lazy = lazy;
    if (R->st.n_big) {
        avk_emu::Wave w;
        w.stack_bytes = 256 * 1024;
        std::vector<char> stacks(64 * w.stack_bytes + 64);
        w.stacks = stacks.data();
        for (uint32_t item = 0; item < R->st.n_big; ++item) {
            DpWaveTask t{&a, item};
            avk_emu::run_wave(&w, dp_wave_main, &t);
        }
    }
    return 0;
}
dpk::DpOpts dp_opts_of(uint64_t lds_bytes, uint32_t lds_ed_cap, uint64_t lds2_bytes, uint32_t lds2_ed_cap, uint32_t solo_min_variants, bool pairs, uint64_t lane_min_regions,
                       uint64_t lane_min_batch = 0, uint32_t lane_max_est = 15) {
    dpk::DpOpts o;
    memset(&o, 0, sizeof(o));
    o.tier0_bytes = avk::bulk_slice_bytes(lds_bytes), o.tier0_ed_cap = lds_ed_cap, o.tier1_bytes = lds2_bytes, o.tier1_ed_cap = lds2_ed_cap;
    o.solo_min_variants = pairs && !g_pair_classes ? 0u : solo_min_variants, o.max_branch = 50;
    o.class_c_nodes_x2 = getenv("AVK_EMU_CLASS_C") ? (uint32_t)atoi(getenv("AVK_EMU_CLASS_C")) : 12u;
    o.lane_min_regions = lane_min_regions, o.lane_max_calls = AVK_FAST_MAXV, o.lane_min_batch = lane_min_batch, o.lane_max_est = lane_max_est;
    o.stripe_w = g_stripe_w;
    o.lane_pairs = g_lane_pairs ? 1u : 0u;
    o.head_est = 1, o.het_min = AVK_HET_SEARCH_MIN;
    o.class_c_below = g_class_c_below;
    return o;
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
    const bool devpack = g_device_pack != 0 && !want_seq; /* (the library lays the sequence slots out itself; this harness writes them straight to the caller's) */
    DpResult dpr;
    int rc = 0;
    if (devpack) {
        rc = dp_run(batch, base, lens, dp_opts_of(lds_bytes, lds_ed_cap, lds2_bytes, lds2_ed_cap, solo_min_variants, mode == 1, g_lane_kernel ? 0ull : 0xFFFFFFFFull), mode == 1, &dpr,
                    /* lazily, exactly when run_internal's lane launches will run (use_fast below) */
                    g_device_pack == 2 && cfg->max_branch_factor != 0 && g_lane_kernel && !cfg->enable_exact_shortcut && !cfg->enable_sequences && lds_bytes > 0 &&
                        !(lds2_bytes > 0 && lds2_overflow_pass) && ws_bytes > 0);
        err = dpr.err;
    } else
        rc = avk::pack_batch(batch, base, lens, want_seq ? out->seq_off : nullptr, want_seq ? out->seq_stride : nullptr, &pb, &err, 0, 15, g_lane_pairs != 0 && mode == 0);
    if (rc) {
        fprintf(stderr, "emu pack error: %s\n", err.c_str());
        return rc;
    }
    const uint64_t n = batch->n_regions, nv = devpack ? dpr.st.total_v : pb.variants.size();
    std::vector<uint32_t> rout(n * 4 + 4, 0), vout(nv + 1, 0);
    std::vector<uint32_t> gm(out->group_metrics ? n * AVK_N_GROUPS * AVK_N_FIELDS : 0);
    std::vector<uint64_t> partials((size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES, 0), tally(AVK_TALLY_STRIDE, 0);
    std::vector<uint32_t> lists[4] = {std::vector<uint32_t>(n + 1), std::vector<uint32_t>(n + 1), std::vector<uint32_t>(n + 1), std::vector<uint32_t>(n + 1)};
    std::vector<uint32_t> counters_v(1280, 0);
    uint32_t *counters = counters_v.data();

    AvkKernelArgs a;
    memset(&a, 0, sizeof(a));
    a.regions = pb.regions.data();
    a.blob = devpack ? dpr.blob.data() : pb.blob.data();
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
    if (mode == 1 && !devpack) { /* the pre-checks of avk_optimize_pairs_batch (aardvark_amd/csrc/avk_host.hip); dp_region applies them itself */
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
    a.tier[2].ed_cap = g_hbm_ed_cap ? (g_hbm_ed_cap | AVK_CAP_BOUND_ONLY) : 0u;
    a.tier[3].ws_bytes = big_ws_bytes;
    a.tier[3].ed_cap = 0;
    a.region_out = rout.data();
    a.group_metrics = out->group_metrics ? gm.data() : nullptr;
    a.var_out = vout.data();
    a.seq_bytes = want_seq ? out->seq_bytes : nullptr;
    a.seq_len = want_seq ? out->seq_len : nullptr;
    a.tally = partials.data();
    std::vector<uint32_t> bp_off_host, bp_dev; /* compact BASEPAIR groups (avk_result_batch::bp_groups): offsets as the packers count them, written straight to the caller's array */
    const bool bp_packed = out->bp_packed && out->bp_spilled && out->bp_groups && !out->bp_off && mode == 0 && devpack; /* the packed form: dp_unpack makes it from the kernels' groups */
    if (bp_packed) {
        bp_dev.assign(4 * ((size_t)dpr.bp_off[n] + 1), 0);
        a.bp_off = dpr.bp_off.data();
        a.bp_out = bp_dev.data();
    }
    if (out->bp_off && out->bp_groups && mode == 0) {
        if (devpack) memcpy(out->bp_off, dpr.bp_off.data(), (n + 1) * sizeof(uint32_t));
        else {
            out->bp_off[0] = 0;
            for (uint64_t r = 0; r < n; ++r) {
                const uint32_t ps = pb.regions[r].pre_status;
                out->bp_off[r + 1] = out->bp_off[r] + ((ps & 0xFFFFu) ? 0u : 1u + (uint32_t)__builtin_popcount(ps >> 16));
            }
        }
        a.bp_off = out->bp_off;
        a.bp_out = out->bp_groups;
    }

    if (cfg->max_branch_factor == 0) { /* query_optimizer.rs:177 */
        for (uint64_t r = 0; r < n; ++r) {
            AvkDevRegion &dr = devpack ? dpr.regions[r] : pb.regions[r];
            if (!(dr.pre_status & 0xFFFFu)) dr.pre_status = AVK_ST_BRANCH_FACTOR;
        }
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

    /* a launch of avk_wide.inl (run_internal: avk_wide_kernel / avk_wide_kernel_lazy): `waves` one-wave workgroups on the list of `w` */
    auto run_wide = [&](const AvkKernelArgs &w, uint32_t waves, uint32_t skip_static = 0, uint32_t lds_bytes_w = 0) {
        if (!lds_bytes_w) lds_bytes_w = g_wide_lds_bytes;
        std::atomic<uint32_t> next(0);
        auto worker = [&]() {
            avk_emu::Wave wv;
            wv.stack_bytes = 256 * 1024;
            std::vector<char> stacks(64 * wv.stack_bytes + 64);
            wv.stacks = stacks.data();
            std::vector<uint32_t> ldsbuf(lds_bytes_w / 4 + 64);
            for (;;) {
                const uint32_t wid = next.fetch_add(1);
                if (wid >= waves) break;
                for (auto &x : ldsbuf) x = 0xA5A5A5A5u; /* LDS is not zeroed */
                WideTask t;
                t.args = &w, t.wa.lds_words = lds_bytes_w / 4, t.wa.skip_static = skip_static, t.wave_id = wid, t.lds = ldsbuf.data();
                avk_emu::run_wave(&wv, wide_kernel_main, &t);
            }
        };
        std::vector<std::thread> ts;
        for (int i = 0; i < (threads < 1 ? 1 : threads); ++i) ts.emplace_back(worker);
        for (auto &t : ts) t.join();
    };
    std::vector<uint32_t> wide_left_c(n + 1), wide_left_3(n + 1), wide_left_l(n + 1), wide_left_c2(n + 1); /* what the wide launches could not take (run_internal: d_overflow5 / d_overflow6) */
    const bool use_wide = g_wide_kernel && lds_bytes > 0 && !(lds2_bytes > 0 && lds2_overflow_pass) && ws_bytes > 0 && !cfg->enable_sequences && !cfg->enable_exact_shortcut && n;

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
    avk::WorkPlan plan;
    avk::PodVec<AvkDevRegion> sorted;
    if (devpack) { /* the packer wrote the records in work order and made the plan */
        plan.n_hbm = dpr.st.n_hbm, plan.n_hard = dpr.st.n_hard, plan.n_fast_total = dpr.st.n_fast_total;
        for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) plan.n_fast[fc] = dpr.st.n_fast[fc], plan.n_fast_heavy[fc] = dpr.st.n_fast_heavy[fc], plan.fast_base[fc] = dpr.st.fast_base[fc];
    } else {
        plan = avk::plan_work_order(pb, avk::bulk_slice_bytes(lds_bytes), lds_ed_cap, lds2_bytes, lds2_ed_cap, mode == 1 && !g_pair_classes ? 0u : solo_min_variants, 50, &order,
                                    getenv("AVK_EMU_CLASS_C") ? (uint32_t)atoi(getenv("AVK_EMU_CLASS_C")) : 12u,
                                    g_lane_kernel ? 0ull : 0xFFFFFFFFull /* as upload_internal does with the option lane_kernel off */, AVK_FAST_MAXV, 0, g_stripe_w, 1, AVK_HET_SEARCH_MIN, g_class_c_below);
        sorted = avk::regions_in_work_order(pb, order); /* the records go in work order */
    }
    a.regions = devpack ? dpr.regions.data() : sorted.data();
    if (devpack && dpr.lazy) { /* run_internal: the waves that take handed-back regions write their records themselves */
        a.lazy_dp = &dpr.args;
        a.lazy_from = (uint32_t)n - dpr.st.n_fast_total;
    }
    /* the lane-per-region launches of run_internal (aardvark_amd/csrc/avk_host.hip): fast segments first, leftovers to the list the
     * first HBM pass reads */
    const int fast_list = launch[1] ? 1 : 0;
    const bool use_fast = g_lane_kernel && launch[0] && !launch[1] && launch[2] && !launch[3] && plan.n_fast_total && !cfg->enable_sequences && !cfg->enable_exact_shortcut && n;
    (void)fast_list;
    const uint32_t n_fast = use_fast ? plan.n_fast_total : 0u;
    g_lane_solved = 0;
    g_quad_solved = 0;
    g_last_pair_regions = 0;
    if (use_fast) {
        uint64_t word_base[AVK_FAST_CLASSES];
        uint32_t n_tiles[AVK_FAST_CLASSES];
        avk::PodVec<uint32_t> fast;
        if (devpack) {
            fast.resize(dpr.fast.size());
            memcpy(fast.data(), dpr.fast.data(), dpr.fast.size() * sizeof(uint32_t));
            for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) word_base[fc] = dpr.st.fast_word_base[fc], n_tiles[fc] = dpr.st.fast_tiles[fc];
        } else
            fast = avk::build_fast_records(pb, order, plan, word_base, n_tiles);
        AvkKernelArgs f = a;
        f.overflow_list = lists[2].data(); /* the DEFERRED list: an LDS pass of the wave-per-region code after the bulk */
        f.overflow_count = counters + 1024 + 32;
        for (int fc = AVK_FAST_CLASSES - 1; fc >= 0; --fc) {
            if (!n_tiles[fc]) continue;
            if (fc == AVK_FAST_PAIR && mode == 0 && g_lane_pairs) { /* run_internal: the table (ensure_pair_table), then the lookups */
                namespace pr = avk::pairs;
                g_last_pair_regions = plan.n_fast[fc];
                static pr::PairTable tab; /* (one emulated call at a time fills and reads it) */
                memset(&tab, 0xFF, sizeof(tab));
                std::vector<uint32_t> aux(896 + 2 * AVK_TALLY_STRIDE, 0);
                pr::pair_probe_records(aux.data(), aux.data() + 768);
                for (uint32_t k = 0; k <= pr::N_SIG; ++k) aux[792 + k] = 2 * k;
                AvkKernelArgs pf;
                memset(&pf, 0, sizeof(pf));
                pf.ref_2bit = aux.data() + 768, pf.ref_exc = aux.data() + 784;
                pf.n_regions = pr::N_SIG, pf.max_branch_factor = cfg->max_branch_factor;
                pf.region_out = &tab.region[0][0], pf.var_out = &tab.var[0][0], pf.group_metrics = &tab.gm[0][0];
                pf.bp_off = aux.data() + 792, pf.bp_out = &tab.bp[0][0];
                pf.tally = (uint64_t *)(aux.data() + 896);
                pf.overflow_list = aux.data() + 816, pf.overflow_count = aux.data() + 813;
                avk::lane::LaneArgs pla;
                memset(&pla, 0, sizeof(pla));
                pla.recs = aux.data(), pla.rec_words = AVK_FAST_WORDS_OF(1), pla.n_tiles = 1, pla.tile_counter = aux.data() + 812;
                pla.W = AVK_FAST_CLASS[0].W, pla.nm = 2, pla.ed_max = AVK_FAST_CLASS[0].ed_max, pla.qcap = AVK_FAST_CLASS[0].qcap, pla.lanes_log2 = 6, pla.max_nodes = 250;
                {
                    avk_emu::Wave w;
                    w.stack_bytes = 256 * 1024;
                    std::vector<char> stacks(64 * w.stack_bytes + 64);
                    w.stacks = stacks.data();
                    std::vector<uint32_t> lds(((size_t)avk::lane::lane_rows(pla.W, pla.nm, pla.ed_max, pla.qcap, pla.pool) << 6) + 64, 0xA5A5A5A5u), tl(288, 0);
                    LaneTask t;
                    t.args = &pf, t.la = &pla, t.wave_id = 0, t.lds = lds.data(), t.tally = tl.data();
                    avk_emu::run_wave(&w, lane_kernel_main, &t);
                }
                pr::PairArgs pa;
                pa.recs = fast.data() + word_base[fc], pa.n_tiles = n_tiles[fc], pa.gen_base = plan.fast_base[fc], pa.tab = &tab, pa.tile_counter = counters + 1220 + fc;
                const int nthr = threads < 1 ? 1 : threads;
                std::vector<uint64_t> sums((size_t)nthr * AVK_TALLY_STRIDE, 0);
                auto worker = [&](int tid) {
                    avk_emu::Wave w;
                    w.stack_bytes = 256 * 1024;
                    std::vector<char> stacks(64 * w.stack_bytes + 64);
                    w.stacks = stacks.data();
                    PairTask t;
                    t.args = &f, t.pa = &pa, t.part = sums.data() + (size_t)tid * AVK_TALLY_STRIDE;
                    avk_emu::run_wave(&w, pair_kernel_main, &t);
                };
                std::vector<std::thread> ts;
                for (int i = 0; i < nthr; ++i) ts.emplace_back(worker, i);
                for (auto &t : ts) t.join();
                for (int i = 0; i < nthr; ++i)
                    for (int k = 0; k < AVK_TALLY_STRIDE; ++k) partials[k] += sums[(size_t)i * AVK_TALLY_STRIDE + k];
                continue;
            }
            /* (the looked-up class in merge mode or with the option off: its records are those of a one-call class) */
            const AvkFastClass &cl = fc == AVK_FAST_PAIR ? AVK_FAST_CLASS[1] : AVK_FAST_CLASS[fc];
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
            const uint32_t pool_heavy = g_lane_pool < 0 ? avk::lane::lane_pool_default(la.nm) : (uint32_t)g_lane_pool; /* run_internal's rule */
            la.pool = g_lane_pool < 0 ? (cl.maxv > 2 ? pool_heavy : 0u) : (uint32_t)g_lane_pool;
            AvkKernelArgs f3 = f; /* run_internal: the three-call class hands back to a list of its own, solved by an HBM-tier launch right behind it */
            f3.overflow_list = lists[3].data();
            f3.overflow_count = counters + 1104;
            auto launch = [&](const avk::lane::LaneArgs &la) {
                const bool quad = g_lane_quad && la.lanes_log2 <= 4; /* run_internal's rule */
                const uint32_t rows = quad ? avk::quad::quad_rows(la.W, la.nm, la.ed_max, la.qcap, la.pool) : avk::lane::lane_rows(la.W, la.nm, la.ed_max, la.qcap, la.pool);
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
                        avk_emu::run_wave(&w, quad ? quad_kernel_main : lane_kernel_main, &t);
                        for (int l = 0; l < 64; ++l) {
                            sm[AVK_TALLY_SOLVED] += t.n_ok[l];
                            sm[AVK_TALLY_ERRORS] += t.n_err[l];
                            sm[AVK_TALLY_LANE_SOLVED] += t.n_ok[l] + t.n_err[l];
                            if (quad) __atomic_fetch_add(&g_quad_solved, (uint64_t)(t.n_ok[l] + t.n_err[l]), __ATOMIC_RELAXED);
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
                hd.pool = pool_heavy;
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
                if (use_wide) { /* ... and the launch of avk_wide.inl ahead of it */
                    AvkKernelArgs w = a;
                    w.work_base = 0, w.n_work = 0, w.work_counter = counters + 1248, w.overflow_list = wide_left_3.data(), w.overflow_count = counters + 1252;
                    run_wide(w, n_waves ? n_waves : 1);
                    a.work_list = wide_left_3.data();
                    a.n_work_dev = counters + 1252;
                }
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
                    if (use_wide) { /* run_internal: class C through avk_wide.inl first, the HBM solo launch takes what is left */
                        AvkKernelArgs w = a;
                        w.work_list = nullptr, w.n_work_dev = nullptr, w.work_base = 0, w.n_work = n_c, w.work_counter = counters + 1240;
                        w.overflow_list = wide_left_c.data(), w.overflow_count = counters + 1244;
                        run_wide(w, n_waves ? n_waves : 1, 1);
                        { /* ... beside the HBM-tier launch for the records that are not the wide kernel's by what they say themselves (AvkKernelArgs::only_not_wide) */
                            AvkKernelArgs keep2 = a;
                            a.pass_tier = 2, a.only_not_wide = 1, a.work_list = nullptr, a.n_work_dev = nullptr, a.work_base = 0, a.n_work = n_c, a.work_counter = counters + 1256;
                            a.static_pct = 0, a.n_shards = 1, a.claim = 4, a.high_priority = 1, a.overflow_list = nullptr, a.overflow_count = nullptr;
                            a.big_ws = big_slots ? big_slices.data() : nullptr, a.big_busy = counters + 1088, a.big_slots = big_slots;
                            run_pass(n_waves / 4 ? n_waves / 4 : 1, ws_bytes, 0);
                            a = keep2;
                        }
                        a.work_list = wide_left_c.data();
                        a.n_work_dev = counters + 1244;
                        if (g_wide_lds_bytes < 64u * 1024u) { /* run_internal: what it handed over once more with the LDS of a whole workgroup (option wide_retry_lds_bytes) */
                            AvkKernelArgs w2 = w;
                            w2.work_list = wide_left_c.data(), w2.n_work_dev = counters + 1244, w2.work_base = 0, w2.n_work = 0, w2.work_counter = counters + 1268;
                            w2.overflow_list = wide_left_c2.data(), w2.overflow_count = counters + 1272;
                            run_wide(w2, 2, 0, 64u * 1024u);
                            a.work_list = wide_left_c2.data();
                            a.n_work_dev = counters + 1272;
                        }
                        hbm_shared = 0;
                    }
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
                    if (use_wide) { /* run_internal: avk_wide.inl first */
                        AvkKernelArgs w = a;
                        w.work_counter = counters + 1260, w.overflow_list = wide_left_l.data(), w.overflow_count = counters + 1264;
                        run_wide(w, n_waves ? n_waves : 1);
                        a.work_list = wide_left_l.data();
                        a.n_work_dev = counters + 1264;
                    }
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
    if (devpack) { /* download_device_packed: dp_unpack writes the caller's layout */
        dpk::DpOut o;
        memset(&o, 0, sizeof(o));
        o.region_out = rout.data(), o.var_out = vout.data(), o.v_off = dpr.v_off.data(), o.t_off = batch->t_off, o.q_off = batch->q_off, o.t_cnt = batch->t_cnt, o.q_cnt = batch->q_cnt;
        o.n_regions = n, o.n_variants = batch->n_variants, o.mode = mode;
        if (dpr.packed_source) o.t_off = o.q_off = nullptr, o.t_cnt = o.q_cnt = nullptr, o.pk_voff = dpr.pk_voff.data(), o.pk_tc = dpr.pk_tc.data(), o.pk_qc = dpr.pk_qc.data();
        o.status = out->status, o.ed_h1 = out->ed_h1, o.ed_h2 = out->ed_h2, o.n_optima = out->n_optima, o.type_present = out->type_present;
        o.var_expected = out->var_expected, o.var_observed = out->var_observed, o.var_class = out->var_class, o.var_zyg = out->var_zyg;
        o.region_packed = out->region_packed, o.var_packed = out->var_packed;
        if (bp_packed) {
            out->bp_spilled[0] = 0;
            o.bp_off_dev = dpr.bp_off.data(), o.bp_dev = bp_dev.data(), o.bp_packed = out->bp_packed, o.bp_spill = out->bp_groups, o.bp_spill_count = out->bp_spilled;
        }
        for (uint64_t r = 0; r < n; ++r) dpk::dp_unpack(o, r);
    }
    /* copy back in caller order */
    for (uint64_t r = 0; r < n && !devpack; ++r) {
        const uint32_t *w = rout.data() + 4 * r;
        if (out->status) out->status[r] = (int32_t)w[0];
        if (out->ed_h1) out->ed_h1[r] = w[1];
        if (out->ed_h2) out->ed_h2[r] = w[2];
        if (out->n_optima) out->n_optima[r] = w[3] & 0xFFFFu;
        if (out->type_present) out->type_present[r] = (uint16_t)(w[3] >> 16);
        if (out->region_packed) out->region_packed[r] = avk_rp_make(w[0], w[1], w[2], w[3] & 0xFFFFu, w[3] >> 16);
    }
    if (out->group_metrics) memcpy(out->group_metrics, gm.data(), gm.size() * sizeof(uint32_t));
    for (uint64_t v = 0; v < nv && !devpack; ++v) {
        const uint64_t hv = pb.dev2host[v];
        const uint32_t w = vout[v];
        if (out->var_expected) out->var_expected[hv] = (uint8_t)(w & 0xFF);
        if (out->var_observed) out->var_observed[hv] = (uint8_t)((w >> 8) & 0xFF);
        if (out->var_class) out->var_class[hv] = (uint8_t)((w >> 16) & 0xFF);
        if (out->var_zyg) out->var_zyg[hv] = (uint8_t)(w >> 24);
        if (out->var_packed) out->var_packed[hv] = avk_vp_make(w & 0xFF, (w >> 8) & 0xFF, w >> 24);
    }
    if (out->tally) memcpy(out->tally, tally.data(), AVK_TALLY_LEN * sizeof(uint64_t));
    if (tier_counts) memcpy(tier_counts, tally.data() + AVK_TALLY_LEN, 5 * sizeof(uint64_t));
    g_lane_solved = tally[AVK_TALLY_LANE_SOLVED];
    g_wide_solved = tally[AVK_TALLY_WIDE_SOLVED];
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

/* avk_merge_batch's device path on emulated wavefronts: dp_expand_pairs, the pair solve (mode 1), dp_merge_classify */
int emu_merge_batch(const avk_multi_batch *mb, const uint8_t *const *refs, const uint64_t *ref_lens, uint32_t n_contigs, const avk_merge_config *cfg, int32_t *status,
                    uint8_t *classification, uint64_t *members, int threads) {
    const uint32_t k = mb->n_inputs, ppr = k * (k - 1) / 2;
    const uint64_t nm = mb->n_regions, np = nm * ppr;
    std::vector<uint64_t> st(np + 1), en(np + 1), toff(np + 1), qoff(np + 1);
    std::vector<uint32_t> cidx(np + 1), tcnt(np + 1), qcnt(np + 1);
    dpk::DpPairs c;
    memset(&c, 0, sizeof(c));
    c.contig_idx = mb->contig_idx, c.start = mb->start, c.end = mb->end, c.in_off = mb->in_off, c.in_cnt = mb->in_cnt, c.n_multi = nm, c.k = k, c.ppr = ppr;
    c.w_contig = mb->contig_idx ? cidx.data() : nullptr, c.w_t_cnt = tcnt.data(), c.w_q_cnt = qcnt.data(), c.w_start = st.data(), c.w_end = en.data(), c.w_t_off = toff.data(), c.w_q_off = qoff.data();
    for (uint64_t p = 0; p < np; ++p) dpk::dp_expand_pairs(c, p);
    avk_region_batch b;
    memset(&b, 0, sizeof(b));
    b.n_regions = np, b.contig_idx = mb->contig_idx ? cidx.data() : nullptr, b.start = st.data(), b.end = en.data(), b.t_off = toff.data(), b.t_cnt = tcnt.data(), b.q_off = qoff.data(),
    b.q_cnt = qcnt.data(), b.n_variants = mb->n_variants, b.var_pos = mb->var_pos, b.var_type = mb->var_type, b.var_zyg = mb->var_zyg, b.var_raw_space = mb->var_raw_space, b.a0_off = mb->a0_off,
    b.a0_len = mb->a0_len, b.a1_off = mb->a1_off, b.a1_len = mb->a1_len, b.allele_bytes = mb->allele_bytes, b.allele_bytes_len = mb->allele_bytes_len;
    std::vector<int32_t> pst(np + 1);
    std::vector<uint32_t> pex(np + 1);
    avk_compare_config pc;
    pc.max_branch_factor = cfg->max_branch_factor, pc.enable_sequences = 0, pc.enable_exact_shortcut = 0;
    avk_result_batch out;
    memset(&out, 0, sizeof(out));
    out.status = pst.data();
    out.ed_h1 = pex.data();
    const int rc = emu_run(1, &b, refs, ref_lens, n_contigs, &pc, &out, 10 * 1024, 48, 40 * 1024, 48, 1 << 20, 64ull << 20, 8, threads, nullptr, 0, 0, 1);
    if (rc) return rc;
    std::vector<uint32_t> rout(4 * np + 4, 0); /* what the pair solve leaves in region_out: status, exact */
    for (uint64_t p = 0; p < np; ++p) rout[4 * p] = (uint32_t)pst[p], rout[4 * p + 1] = pex[p];
    dpk::DpMerge m;
    memset(&m, 0, sizeof(m));
    m.region_out = rout.data(), m.in_off = mb->in_off, m.in_cnt = mb->in_cnt, m.var_zyg = mb->var_zyg, m.n_multi = nm, m.n_variants = mb->n_variants, m.k = k, m.ppr = ppr;
    m.no_conflict_enabled = cfg->no_conflict_enabled, m.majority_voting_enabled = cfg->majority_voting_enabled, m.conflict_selection = cfg->conflict_selection;
    m.status = status, m.classification = classification, m.members = members;
    for (uint64_t r = 0; r < nm; ++r) dpk::dp_merge_classify(m, r);
    return 0;
}

void emu_set_lane_kernel(int on) { g_lane_kernel = on; }
void emu_set_team(int on) { g_team = on; }
void emu_set_device_pack(int on) { g_device_pack = on; }
void emu_set_packed_source(int on) { g_dp_packed_source = on, g_last_packed_source = 0; }
int emu_last_packed_source() { return g_last_packed_source; }
void emu_set_stripe(uint32_t w) { g_stripe_w = w; }
void emu_set_class_c_below(uint64_t n) { g_class_c_below = n; }
void emu_set_lane_pairs(int on) { g_lane_pairs = on; }
void emu_set_hbm_ed_cap(uint32_t cap) { g_hbm_ed_cap = cap; }
uint64_t emu_last_pair_regions() { return g_last_pair_regions; }

/* The device-side packer against the host-side one on the same batch: every region record, every blob, the plan, the work order and the fast
 * records must be identical (the blob arena may be laid out differently: blobs are compared by content).  Returns 0, or 1 with the first
 * difference in `msg`.  lane_min_regions / lane_min_batch / lane_max_est as the context options of the same names. */
int emu_devpack_compare(const avk_region_batch *batch, const uint64_t *ref_lens, uint32_t n_contigs, int pairs_mode, uint64_t lane_min_regions, uint64_t lane_min_batch,
                        uint32_t lane_max_est, uint32_t solo_min_variants, char *msg, size_t msg_len) {
    std::vector<uint64_t> base(n_contigs), lens(n_contigs);
    uint64_t total = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) base[c] = total, lens[c] = ref_lens[c], total += ref_lens[c];
    auto say = [&](const char *fmt, auto... args) {
        snprintf(msg, msg_len, fmt, args...);
        return 1;
    };
    const uint64_t n = batch->n_regions;
    /* host side, as upload_internal does it */
    std::vector<uint64_t> seq_off(n);
    std::vector<uint32_t> seq_stride(n);
    uint64_t seq_total = 0;
    for (uint64_t r = 0; r < n; ++r) seq_stride[r] = avk::seq_stride_of(batch, r);
    for (uint64_t r = 0; r < n; ++r) seq_off[r] = seq_total, seq_total += 5ull * seq_stride[r];
    avk::PackedBatch pb;
    std::string err;
    const int rc_h = avk::pack_batch(batch, base, lens, seq_off.data(), seq_stride.data(), &pb, &err, 0, lane_max_est, g_lane_pairs != 0 && !pairs_mode);
    const uint64_t lds_bytes = 10 * 1024, lds2_bytes = 40 * 1024;
    DpResult R;
    const int rc_d = dp_run(batch, base, lens, dp_opts_of(lds_bytes, 48, lds2_bytes, 48, solo_min_variants, pairs_mode != 0, lane_min_regions, lane_min_batch, lane_max_est), pairs_mode != 0, &R);
    if (rc_h != rc_d) return say("return codes differ: host %d (%s), device %d (%s)", rc_h, err.c_str(), rc_d, R.err.c_str());
    if (rc_h) return err == R.err ? 0 : say("error texts differ: host '%s', device '%s'", err.c_str(), R.err.c_str());
    if (pairs_mode)
        for (uint64_t r = 0; r < n; ++r) {
            AvkDevRegion &dr = pb.regions[r];
            if ((dr.pre_status & 0xFFFFu) == AVK_ST_INVALID_INPUT) continue;
            if (pb.zyg_flags[r] & 1) dr.pre_status = AVK_ST_BAD_ZYGOSITY;
            else if (pb.delta_t[r] != pb.delta_q[r]) dr.pre_status = AVK_PRE_SKIP_OK;
            else if (pb.zyg_flags[r] & 2) dr.pre_status = AVK_ST_BAD_ZYGOSITY;
            else dr.pre_status = 0;
        }
    std::vector<uint32_t> order;
    const avk::WorkPlan plan = avk::plan_work_order(pb, avk::bulk_slice_bytes(lds_bytes), 48, lds2_bytes, 48, pairs_mode && !g_pair_classes ? 0u : solo_min_variants, 50, &order, 12, lane_min_regions,
                                                    AVK_FAST_MAXV, lane_min_batch, g_stripe_w, 1, AVK_HET_SEARCH_MIN, g_class_c_below);
    if (pb.variants.size() != R.st.total_v) return say("per-call output words: host %zu, device %llu", pb.variants.size(), (unsigned long long)R.st.total_v);
    if (seq_total != R.st.total_seq) return say("sequence bytes: host %llu, device %llu", (unsigned long long)seq_total, (unsigned long long)R.st.total_seq);
    if (plan.n_hbm != R.st.n_hbm || plan.n_hard != R.st.n_hard || plan.n_fast_total != R.st.n_fast_total)
        return say("plan: host C %u B %u lanes %u, device C %u B %u lanes %u", plan.n_hbm, plan.n_hard, plan.n_fast_total, R.st.n_hbm, R.st.n_hard, R.st.n_fast_total);
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc)
        if (plan.n_fast[fc] != R.st.n_fast[fc] || plan.n_fast_heavy[fc] != R.st.n_fast_heavy[fc] || (plan.n_fast[fc] && plan.fast_base[fc] != R.st.fast_base[fc]))
            return say("lane class %d: host n %u heavy %u base %u, device n %u heavy %u base %u", fc, plan.n_fast[fc], plan.n_fast_heavy[fc], plan.fast_base[fc], R.st.n_fast[fc],
                       R.st.n_fast_heavy[fc], R.st.fast_base[fc]);
    for (uint64_t k = 0; k < n; ++k)
        if (order[k] != R.order[k]) return say("work order differs at %llu: host region %u, device region %u", (unsigned long long)k, order[k], R.order[k]);
    for (uint64_t k = 0; k < n; ++k) {
        const uint32_t r = order[k];
        AvkDevRegion h = pb.regions[r];
        h.orig = r;
        const AvkDevRegion &d = R.regions[k];
        if (h.ref_off != d.ref_off || h.len != d.len || h.v_off != d.v_off || h.t_cnt != d.t_cnt || h.q_cnt != d.q_cnt || h.pre_status != d.pre_status || h.seq_stride != d.seq_stride ||
            h.seq_off != d.seq_off || h.blob_bytes != d.blob_bytes || h.alle_bytes != d.alle_bytes || h.grow != d.grow || h.orig != d.orig || h.ed_bound != d.ed_bound)
            return say("record of region %u differs: host ref %llu len %u v_off %u t %u q %u pre %x stride %u seq_off %llu blob %u alle %u grow %u ed %u | device ref %llu len %u v_off %u t %u q %u pre %x "
                       "stride %u seq_off %llu blob %u alle %u grow %u ed %u",
                       r, (unsigned long long)h.ref_off, h.len, h.v_off, h.t_cnt, h.q_cnt, h.pre_status, h.seq_stride, (unsigned long long)h.seq_off, h.blob_bytes, h.alle_bytes, h.grow, h.ed_bound,
                       (unsigned long long)d.ref_off, d.len, d.v_off, d.t_cnt, d.q_cnt, d.pre_status, d.seq_stride, (unsigned long long)d.seq_off, d.blob_bytes, d.alle_bytes, d.grow, d.ed_bound);
        if (h.blob_bytes && memcmp(pb.blob.data() + 2ull * h.blob_off, R.blob.data() + 2ull * d.blob_off, h.blob_bytes) != 0) {
            const uint8_t *x = (const uint8_t *)(pb.blob.data() + 2ull * h.blob_off), *y = (const uint8_t *)(R.blob.data() + 2ull * d.blob_off);
            uint32_t at = 0;
            while (x[at] == y[at]) ++at;
            return say("blob of region %u (%u + %u calls, %u bytes) differs at byte %u: host %02x, device %02x", r, h.t_cnt, h.q_cnt, h.blob_bytes, at, x[at], y[at]);
        }
    }
    if (plan.n_fast_total) {
        uint64_t word_base[AVK_FAST_CLASSES];
        uint32_t n_tiles[AVK_FAST_CLASSES];
        const avk::PodVec<uint32_t> fast = avk::build_fast_records(pb, order, plan, word_base, n_tiles);
        for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) {
            if (n_tiles[fc] != R.st.fast_tiles[fc] || (n_tiles[fc] && word_base[fc] != R.st.fast_word_base[fc]))
                return say("fast records of class %d: host %u tiles at word %llu, device %u tiles at word %llu", fc, n_tiles[fc], (unsigned long long)word_base[fc], R.st.fast_tiles[fc],
                           (unsigned long long)R.st.fast_word_base[fc]);
            const uint64_t words = (uint64_t)n_tiles[fc] * AVK_FAST_WORDS_OF(AVK_FAST_CLASS[fc].maxv) * 64u;
            for (uint64_t w = 0; w < words; ++w)
                if (fast[word_base[fc] + w] != R.fast[word_base[fc] + w])
                    return say("fast record word %llu of class %d differs: host %08x, device %08x", (unsigned long long)w, fc, fast[word_base[fc] + w], R.fast[word_base[fc] + w]);
        }
    }
    return 0;
}

/* dp_myers64 / dp_variant's alt_ed against the host's edit distance */
uint32_t emu_devpack_alt_ed(const uint8_t *a0, uint32_t l0, const uint8_t *a1, uint32_t l1, int *pending) {
    std::vector<uint8_t> bytes(l0 + l1 + 1);
    memcpy(bytes.data(), a0, l0);
    memcpy(bytes.data() + l0, a1, l1);
    const uint64_t o0 = 0, o1 = l0;
    dpk::DpArgs a;
    memset(&a, 0, sizeof(a));
    a.in.a0_off = &o0, a.in.a1_off = &o1, a.in.a0_len = &l0, a.in.a1_len = &l1, a.in.alleles = bytes.data(), a.in.alleles_len = l0 + l1, a.in.n_variants = 1;
    dpk::DpVarInfo vi;
    dpk::DpState st;
    memset(&st, 0, sizeof(st));
    uint32_t pend[2];
    a.vinfo = &vi, a.st = &st, a.pending = pend;
    dpk::dp_variant(a, 0);
    *pending = (vi.flags & dpk::DP_VF_PENDING) ? 1 : 0;
    return vi.alt_ed;
}
#ifdef AVK_LANE_STATS
void emu_lane_work(uint32_t *per_region) { avk::lane::g_lane_work = per_region; }
void emu_lane_comp(uint32_t *per_region_x8) { avk::lane::g_lane_comp = per_region_x8; }
void emu_lane_stats(uint64_t *out, int reset) {
    for (int i = 0; i < 32; ++i) {
        out[i] = avk::lane::g_lane_stats[i];
        if (reset) avk::lane::g_lane_stats[i] = 0;
    }
}
#endif
uint64_t emu_last_lane_solved(void) { return g_lane_solved; }
uint64_t emu_last_wide_solved(void) { return g_wide_solved; }
void emu_set_wide_kernel(int on) { g_wide_kernel = on; }
#ifdef AVK_WIDE_STATS
void emu_wide_defer_stats(uint64_t *out, int reset) {
    for (int i = 0; i < 64; ++i) {
        out[i] = avk::wide::g_wide_defer[i] | ((uint64_t)avk::wide::g_wide_defer_region[i] << 32);
        if (reset) avk::wide::g_wide_defer[i] = 0;
    }
}
#endif
void emu_set_wide_lds_bytes(uint32_t bytes) { g_wide_lds_bytes = bytes; }
void emu_set_lane_node_cap(int cap) { g_lane_node_cap = (uint32_t)cap; }
void emu_set_lane_pool(int slots) { g_lane_pool = slots; }
void emu_set_lane_quad(int on) { g_lane_quad = on; }
uint64_t emu_last_quad_solved(void) { return g_quad_solved; }
void emu_set_pair_classes(int on) { g_pair_classes = on; }
void emu_set_lane_head_width(int w) { g_lane_head_width = (uint32_t)w; }
static uint32_t width_log2(int w) { return w <= 4 ? 2u : (w <= 8 ? 3u : (w <= 16 ? 4u : (w <= 32 ? 5u : 6u))); }
void emu_set_lane_width(int one, int two) {
    g_lane_width_log2[0] = width_log2(one);
    g_lane_width_log2[1] = width_log2(two);
}
void emu_set_lane_width_three(int three) { g_lane_width_log2[2] = width_log2(three); }

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
    int rc = emu_run(1, batch, refs, ref_lens, n_contigs, &cfg, &out, 10 * 1024, 48, 40 * 1024, 48, 1 << 20, 64ull << 20, 8, threads, nullptr, g_pair_classes ? 5u : 0u, 0, 1); /* (context option
                                                                                                     pair_classes: pair batches plan their classes C and B with solo_min_variants 5) */
    if (rc) return rc;
    for (uint64_t r = 0; r < batch->n_regions; ++r) is_exact_match[r] = status[r] == 0 && ed1[r] ? 1 : 0;
    return 0;
}

} /* extern "C" */
