/*
 * avk_wave.h — the wavefront-level primitives the solver kernels are written against.
 *
 * The solver (avk_solver.inl) maps ONE region to ONE 64-lane wavefront: control flow is
 * wave-uniform, data-parallel work (byte compares along wavefront diagonals, sequence appends,
 * node copies, priority-queue scans) is spread over the 64 lanes and combined with ballot /
 * cross-lane reductions.  Everything cross-lane goes through the few functions below.
 *
 * Two bindings:
 *   - hipcc (gfx950): the real thing — __ballot, DPP row-shift/row-broadcast reductions,
 *     readfirstlane for uniform values (ds_bpermute shuffles only where a true permute is needed).
 *   - AVK_EMU (g++, tests/emu/): a kernel-logic emulator that runs the 64 lanes of a wave as 64
 *     cooperative fibers and rendezvouses at every primitive.  It exists so that the SAME kernel
 *     source can be executed and fuzzed against the oracle in the GPU-less dev container
 *     (`-m "not gpu"` tests).  It is test infrastructure: it is never compiled into
 *     libaardvark_amd.so and is not a fallback.
 */
#ifndef AVK_WAVE_H
#define AVK_WAVE_H

#include <stdint.h>

#ifdef AVK_EMU
/* ------------------------------------------------------------------------------ emulator */
#define AVK_DEV static inline
#define AVK_DEV_MEMBER inline /* member functions of device structs */
#define AVK_DEV_M inline /* member functions */
#define AVK_DEV_NOINLINE static
#define AVK_HD static inline /* plain functions the host code calls as well */
namespace avk_emu {
/* deposits `v` for this lane, runs the other lanes up to the same call site, returns the 64
 * deposited values.  `site` must be identical on all lanes (checked: catches divergent use). */
const uint64_t *gather(uint64_t v, uint32_t site);
int lane();
void yield(); /* lets the other emulated waves (OS threads) run */
} // namespace avk_emu
#define AVK_SITE ((uint32_t)__LINE__)

struct alignas(16) avk_u4 {
    uint32_t x, y, z, w;
};
AVK_DEV int wv_lane() { return avk_emu::lane(); }
AVK_DEV uint64_t wv_ballot_(bool p, uint32_t site) {
    const uint64_t *g = avk_emu::gather(p ? 1 : 0, site);
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i) m |= (g[i] & 1ull) << i;
    return m;
}
AVK_DEV uint32_t wv_shfl_(uint32_t v, int src, uint32_t site) { return (uint32_t)avk_emu::gather(v, site)[src & 63]; }
AVK_DEV uint32_t wv_uni_(uint32_t v, uint32_t site) { /* value must already be wave-uniform */
    const uint64_t *g = avk_emu::gather(v, site | 0x80000000u);
    return (uint32_t)g[0];
}
AVK_DEV uint32_t wv_max_u32_(uint32_t v, uint32_t site) {
    const uint64_t *g = avk_emu::gather(v, site);
    uint32_t m = 0;
    for (int i = 0; i < 64; ++i) m = (uint32_t)g[i] > m ? (uint32_t)g[i] : m;
    return m;
}
AVK_DEV uint32_t wv_min_u32_(uint32_t v, uint32_t site) {
    const uint64_t *g = avk_emu::gather(v, site);
    uint32_t m = 0xFFFFFFFFu;
    for (int i = 0; i < 64; ++i) m = (uint32_t)g[i] < m ? (uint32_t)g[i] : m;
    return m;
}
AVK_DEV uint32_t wv_sum_u32_(uint32_t v, uint32_t site) {
    const uint64_t *g = avk_emu::gather(v, site);
    uint32_t s = 0;
    for (int i = 0; i < 64; ++i) s += (uint32_t)g[i];
    return s;
}
AVK_DEV uint64_t wv_min_u64_(uint64_t v, uint32_t site) {
    const uint64_t *g = avk_emu::gather(v, site);
    uint64_t m = ~0ull;
    for (int i = 0; i < 64; ++i) m = g[i] < m ? g[i] : m;
    return m;
}
AVK_DEV uint32_t wv_readlane_(uint32_t v, uint32_t src, uint32_t site) { return (uint32_t)avk_emu::gather(v, site)[src & 63]; }
AVK_DEV void wv_sync_(uint32_t site) { (void)avk_emu::gather(0, site); }
#define wv_ballot(p) wv_ballot_((p), AVK_SITE)
#define wv_shfl(v, src) wv_shfl_((v), (src), AVK_SITE)
#define wv_uni(v) wv_uni_((v), AVK_SITE)
#define wv_uni64(v) (((uint64_t)wv_uni((uint32_t)((uint64_t)(v) >> 32)) << 32) | (uint64_t)wv_uni((uint32_t)(uint64_t)(v)))
#define wv_max_u32(v) wv_max_u32_((v), AVK_SITE)
#define wv_min_u32(v) wv_min_u32_((v), AVK_SITE)
#define wv_sum_u32(v) wv_sum_u32_((v), AVK_SITE)
#define wv_min_u64(v) wv_min_u64_((v), AVK_SITE)
#define wv_readlane(v, src) wv_readlane_((v), (src), AVK_SITE)
/* the value of the lane below / above (lane 0 / lane 63 keep their own) */
AVK_DEV uint32_t wv_from_below_(uint32_t v, uint32_t site) {
    const int l = avk_emu::lane();
    return (uint32_t)avk_emu::gather(v, site)[l ? l - 1 : 0];
}
AVK_DEV uint32_t wv_from_above_(uint32_t v, uint32_t site) {
    const int l = avk_emu::lane();
    return (uint32_t)avk_emu::gather(v, site)[l < 63 ? l + 1 : 63];
}
#define wv_from_below(v) wv_from_below_((v), AVK_SITE)
#define wv_from_above(v) wv_from_above_((v), AVK_SITE)
#define wv_sync() wv_sync_(AVK_SITE)

/* QUAD primitives (avk_quad.inl: four lanes on one region).  Lanes 4 g .. 4 g + 3 form quad g; control flow around these calls is
 * quad-uniform but NOT wave-uniform (sixteen quads in sixteen places), so the emulator's rendezvous is of the quad's four fibers only. */
namespace avk_emu {
const uint64_t *quad_gather(uint64_t v, uint32_t site); /* the four deposited values of this lane's quad */
}
template <int SRC> AVK_DEV uint32_t qd_bcast_(uint32_t v, uint32_t site) { return (uint32_t)avk_emu::quad_gather(v, site)[SRC]; }
template <int M> AVK_DEV uint32_t qd_xor_(uint32_t v, uint32_t site) {
    const int l = avk_emu::lane() & 3;
    return (uint32_t)avk_emu::quad_gather(v, site)[l ^ M];
}
AVK_DEV void qd_sync_(uint32_t site) { (void)avk_emu::quad_gather(0, site); }
#define qd_bcast(SRC, v) qd_bcast_<SRC>((v), AVK_SITE)
#define qd_xor(M, v) qd_xor_<M>((v), AVK_SITE)
#define qd_sync() qd_sync_(AVK_SITE)

AVK_DEV uint32_t avk_atomic_add_u32(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); } /* waves of a workgroup are OS threads */
AVK_DEV uint32_t avk_atomic_add_u32_global(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
AVK_DEV void avk_atomic_add_u64_global(uint64_t *p, uint64_t v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
AVK_DEV void avk_atomic_or_u32_global(uint32_t *p, uint32_t v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
AVK_DEV void avk_atomic_max_u64_global(uint64_t *p, uint64_t v) {
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
AVK_DEV uint32_t avk_atomic_cas_u32_global(uint32_t *p, uint32_t expect, uint32_t desired) {
    __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return expect;
}
/* many lanes adding to the counters of one table (the emulated lanes are fibers of one thread: a plain add each) */
AVK_DEV void avk_tally_add_u32(uint32_t *table, uint32_t off, uint32_t v) { __atomic_fetch_add(table + off, v, __ATOMIC_RELAXED); }
AVK_DEV uint64_t avk_clock() { return 0; }
/* identity the optimiser cannot see through: keeps per-lane address arithmetic inside the loop it belongs to */
AVK_DEV uint32_t avk_opaque_u32(uint32_t v) { return v; }
AVK_DEV uint32_t avk_ld_agent_u32(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
AVK_DEV void avk_st_agent_u32(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
AVK_DEV void avk_release_agent() { __atomic_thread_fence(__ATOMIC_RELEASE); }
AVK_DEV void avk_acquire_agent() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
AVK_DEV void avk_sleep() { avk_emu::yield(); }
AVK_DEV void avk_sleep_short() { avk_emu::yield(); }
/* workgroup-scope fences around the LDS words below when they publish data in GLOBAL memory to the other waves of the workgroup (the team of avk_solver.inl) */
AVK_DEV void avk_release_wg() { __atomic_thread_fence(__ATOMIC_RELEASE); }
AVK_DEV void avk_acquire_wg() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
/* words shared by the waves of one workgroup (LDS on the device) */
AVK_DEV uint32_t avk_wg_load(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
AVK_DEV void avk_wg_store(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
AVK_DEV uint32_t avk_wg_add(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
AVK_DEV uint32_t avk_wg_cas(uint32_t *p, uint32_t expect, uint32_t desired) {
    __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return expect;
}
AVK_DEV uint32_t avk_wg_xchg(uint32_t *p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_ACQ_REL); }
AVK_DEV int avk_ctz64(uint64_t x) { return __builtin_ctzll(x); }
AVK_DEV int avk_popc64(uint64_t x) { return __builtin_popcountll(x); }

#else
/* -------------------------------------------------------------------------------- gfx950 */
#include <hip/hip_runtime.h>
#define AVK_DEV __device__ __forceinline__
#define AVK_DEV_MEMBER __host__ __device__ __forceinline__
#define AVK_DEV_M __device__ __forceinline__ /* member functions */
#define AVK_DEV_NOINLINE __device__ __noinline__
#define AVK_HD __host__ __device__ inline /* plain functions the host code calls as well */
typedef uint4 avk_u4; /* one 16-byte LDS / global access */

/* The lane index is deliberately opaque to the optimiser (a volatile asm, two VALU instructions per use): as a pure
 * function it and everything derived from it (lane * 28 + 16, lane < 32, ...) is hoisted out of the persistent region loop,
 * where dozens of such invariants outgrow the register file, are spilled in the prologue and reloaded for every region. */
AVK_DEV int wv_lane() {
    unsigned l;
#ifdef AVK_LANE_MBCNT
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
#else
    /* one instruction: the workgroups are one-dimensional and the lanes of a wave are 64 consecutive work-items */
    asm volatile("v_and_b32 %0, 63, %1" : "=v"(l) : "v"(__builtin_amdgcn_workitem_id_x()));
#endif
    return (int)l;
}
AVK_DEV uint64_t wv_ballot(bool p) { return __ballot(p); }
AVK_DEV uint32_t wv_shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
/* value of lane `src` (src must be wave-uniform) as a scalar */
AVK_DEV uint32_t wv_readlane(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }
/* the value of the lane below / above (lane 0 / lane 63 keep their own): one DPP move over the whole wave (wave_shr:1 / wave_shl:1) */
AVK_DEV uint32_t wv_from_below(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false); }
AVK_DEV uint32_t wv_from_above(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false); }
/* a value that is identical on every lane: move it to an SGPR so branches on it are scalar */
AVK_DEV uint32_t wv_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
AVK_DEV uint64_t wv_uni64(uint64_t v) { return ((uint64_t)wv_uni((uint32_t)(v >> 32)) << 32) | (uint64_t)wv_uni((uint32_t)v); }
/* cross-lane reductions on the DPP network (row shifts inside 16 lanes, then row broadcasts):
 * no LDS traffic, a handful of VALU cycles.  `ident` is the neutral element of the operation and
 * fills lanes that have no source.  The result lands in lane 63 and is read back as a scalar. */
#define AVK_DPP_STEP(op, ctrl, rmask)                                                             \
    {                                                                                             \
        const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, ctrl, rmask, 0xf, false); \
        v = op(v, t_);                                                                            \
    }
#define AVK_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
#define AVK_OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define AVK_OP_ADD(a, b) ((a) + (b))
#define AVK_DPP_REDUCE(op)              \
    AVK_DPP_STEP(op, 0x111, 0xf) /* row_shr:1 */  \
    AVK_DPP_STEP(op, 0x112, 0xf) /* row_shr:2 */  \
    AVK_DPP_STEP(op, 0x114, 0xf) /* row_shr:4 */  \
    AVK_DPP_STEP(op, 0x118, 0xf) /* row_shr:8 */  \
    AVK_DPP_STEP(op, 0x142, 0xa) /* row_bcast:15 */ \
    AVK_DPP_STEP(op, 0x143, 0xc) /* row_bcast:31 */ \
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);

AVK_DEV uint32_t wv_max_u32(uint32_t v) {
    const uint32_t ident = 0u;
    AVK_DPP_REDUCE(AVK_OP_MAX)
}
AVK_DEV uint32_t wv_min_u32(uint32_t v) {
    const uint32_t ident = 0xFFFFFFFFu;
    AVK_DPP_REDUCE(AVK_OP_MIN)
}
AVK_DEV uint32_t wv_sum_u32(uint32_t v) {
    const uint32_t ident = 0u;
    AVK_DPP_REDUCE(AVK_OP_ADD)
}
/* 64-bit minimum as two 32-bit reductions: high words first, then the low words of the ties */
AVK_DEV uint64_t wv_min_u64(uint64_t v) {
    const uint32_t hi = wv_min_u32((uint32_t)(v >> 32));
    const uint32_t lo = wv_min_u32((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0xFFFFFFFFu);
    return ((uint64_t)hi << 32) | lo;
}
/* orders this wave's own memory traffic: lanes of one wave communicate through LDS / their
 * private HBM slice, the hardware keeps a wave's DS and VMEM operations in issue order, so only
 * the compiler has to be stopped from moving or caching accesses across this point */
AVK_DEV void wv_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
/* QUAD primitives (avk_quad.inl: four lanes on one region): DPP quad permutes, one VALU move each.  They are used in code that is uniform
 * over the quad but divergent over the wave; the four lanes of a quad are active together there, which is all a quad_perm reads. */
template <int SRC> AVK_DEV uint32_t qd_bcast_(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, SRC * 0x55, 0xf, 0xf, false); }
template <int M> AVK_DEV uint32_t qd_xor_(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, M == 1 ? 0xB1 : (M == 2 ? 0x4E : 0x1B), 0xf, 0xf, false);
}
#define qd_bcast(SRC, v) qd_bcast_<SRC>(v)
#define qd_xor(M, v) qd_xor_<M>(v)
/* lanes of a quad exchange through LDS rows as well: the hardware keeps a wave's DS operations in order, the compiler must */
AVK_DEV void qd_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
AVK_DEV uint32_t avk_atomic_add_u32(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
AVK_DEV uint32_t avk_atomic_add_u32_global(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
AVK_DEV void avk_atomic_add_u64_global(uint64_t *p, uint64_t v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }
AVK_DEV void avk_atomic_or_u32_global(uint32_t *p, uint32_t v) { atomicOr(p, v); }
AVK_DEV void avk_atomic_max_u64_global(uint64_t *p, uint64_t v) { atomicMax((unsigned long long *)p, (unsigned long long)v); }
AVK_DEV uint32_t avk_atomic_cas_u32_global(uint32_t *p, uint32_t expect, uint32_t desired) { return atomicCAS(p, expect, desired); }
/* Many lanes adding to the counters of one LDS table, called from DIVERGENT code by whichever lanes have something to add: lanes that add the SAME value
 * to the SAME counter are combined into one atomic (the first remaining lane's pair is broadcast, the lanes that match it leave together) — a tile of the
 * modal class adds the same ones and twos to the same dozen counters from all 64 lanes, which as 64 separate ds_add on one address serialise
 * (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.70 on that launch). */
AVK_DEV void avk_tally_add_u32(uint32_t *table, uint32_t off, uint32_t v) {
    for (;;) {
        const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off), v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        const bool same = off == off0 && v == v0;
        const unsigned long long m = __ballot(same);
        if (same) {
            if ((unsigned)wv_lane() == (unsigned)(__ffsll((unsigned long long)m) - 1)) atomicAdd(table + off0, v0 * (uint32_t)__popcll(m));
            break;
        }
    }
}
AVK_DEV uint64_t avk_clock() { return __builtin_amdgcn_s_memtime(); }
/* identity the optimiser cannot see through: keeps per-lane address arithmetic inside the loop it belongs to (hoisted
 * out of the persistent region loop it is spilled to scratch in the prologue and reloaded for every region) */
AVK_DEV uint32_t avk_opaque_u32(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
/* device-scope (all XCDs) accesses for words that other workgroups poll: write-through / L1-bypassing forms
 * (cdna_hip_programming.md §6 G16: granule = one aligned word that is its own flag) */
AVK_DEV uint32_t avk_ld_agent_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
AVK_DEV void avk_st_agent_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
AVK_DEV void avk_release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
AVK_DEV void avk_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
AVK_DEV void avk_sleep() { __builtin_amdgcn_s_sleep(32); }
AVK_DEV void avk_sleep_short() { __builtin_amdgcn_s_sleep(2); }
/* workgroup-scope fences around the LDS words below when they publish data in GLOBAL memory to the other waves of the workgroup (the team of avk_solver.inl): the
 * waves of a workgroup share their CU's vector L1, so completing the stores (release) and ordering the later loads (acquire) is all there is to it */
AVK_DEV void avk_release_wg() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
AVK_DEV void avk_acquire_wg() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
/* words shared by the waves of one workgroup (they live in LDS) */
AVK_DEV uint32_t avk_wg_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
AVK_DEV void avk_wg_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
AVK_DEV uint32_t avk_wg_add(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
AVK_DEV uint32_t avk_wg_cas(uint32_t *p, uint32_t expect, uint32_t desired) {
    __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return expect;
}
AVK_DEV uint32_t avk_wg_xchg(uint32_t *p, uint32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
AVK_DEV int avk_ctz64(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }
AVK_DEV int avk_popc64(uint64_t x) { return __popcll(x); }
#endif

#endif /* AVK_WAVE_H */
