/*
 * avk_host.hip — libaardvark_amd.so: the C-ABI of include/aardvark_amd.h on top of the gfx950
 * solver kernels.  Host code here owns device memory, streams and launches; the per-region
 * algorithm lives in avk_solver.inl and runs only on the GPU (there is no CPU path in this
 * library: without a HIP device every entry point fails with AVK_E_HIP).
 *
 * Launch shape: persistent workgroups of independent wavefronts, one wavefront solves one
 * region end to end in its workspace slice.  A step starts up to three launches side by side — the bulk (small LDS
 * slices, caller's stream) and, on two side streams, the solo launches of the regions the upload-time work plan
 * predicted to need a large LDS slice or an HBM slice — followed by one HBM launch for whatever still overflowed
 * (its list length is read on the device, so nothing returns to the host in between) and the tally reduce, which
 * also clears the counters for the next step.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aardvark_amd.h"
#include "avk_pack.h"
#include "avk_solver.inl"
#include "avk_lane.inl"
#include "avk_quad.inl"
#include "avk_wide.inl"
#include "avk_dwfa_script.inl"
#include "avk_devpack.inl"

/* ---------------------------------------------------------------------------------- kernels */
/* LDS passes: regions in the LDS slice of their wavefront (small slices at high occupancy first, then
 * the overflow of that pass with large slices at one workgroup per CU) */
#ifndef AVK_LDS_WAVES_PER_SIMD
#define AVK_LDS_WAVES_PER_SIMD 4
#endif
__global__ void __launch_bounds__(256, AVK_LDS_WAVES_PER_SIMD) avk_region_kernel_lds(AvkKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    const unsigned wave_in_block = threadIdx.x >> 6;
    const unsigned wave_id = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    if (a.high_priority) __builtin_amdgcn_s_setprio(3); /* solo launch: the long searches are the critical path */
    if (a.esc_bytes) { /* the workgroup's tail: control words and tally (AvkKernelArgs::esc_bytes) */
        for (unsigned k = threadIdx.x; k < AVK_WG_TAIL_BYTES / 4; k += blockDim.x) ((unsigned *)(avk_smem + a.esc_bytes))[k] = 0;
        __syncthreads();
    }
    avk::region_worker<true>(a, wave_id, avk_smem + (size_t)wave_in_block * a.tier[a.pass_tier].ws_bytes);
}

/* HBM passes: regions that outgrew the LDS tiers, in the wave's private HBM slice */
__global__ void __launch_bounds__(256, 2) avk_region_kernel_hbm(AvkKernelArgs a) {
    const unsigned wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    avk::region_worker<false>(a, wave_id, (unsigned char *)0);
}

/* The long windows, a workgroup per region (avk_solver.inl "a region on a TEAM of wavefronts"): wave 0 claims regions and runs their searches, its three siblings
 * take the independent pieces it posts — haplotype extensions of a popped node's children, the alignments of the metrics.  Every wave has an HBM slice: the
 * owner's is the region's workspace, a sibling's is its scratch. */
__global__ void __launch_bounds__(256, 2) avk_region_kernel_team(AvkKernelArgs a) {
    __shared__ avk::TeamBox box;
    for (unsigned k = threadIdx.x; k < sizeof(box) / 4; k += blockDim.x) ((uint32_t *)&box)[k] = 0;
    __syncthreads();
    const unsigned w = threadIdx.x >> 6;
    if (a.high_priority) __builtin_amdgcn_s_setprio(3);
    if (w == 0) {
        avk::region_worker<false, false, true>(a, blockIdx.x * 4u, (unsigned char *)0, &box);
        wv_sync();
        if ((threadIdx.x & 63u) == 0) avk_wg_store(&box.quit, 1u);
    } else if (a.team == 1) { /* (2 = the owner alone takes every job: a diagnostic) */
        avk::team_helper(&box, a.hbm_ws + (uint64_t)(blockIdx.x * 4u + w) * a.tier[a.pass_tier].ws_bytes, a.tier[a.pass_tier].ws_bytes, w, 4u);
    }
}

/* the same two for the regions the lanes handed back: device-packed batches (avk_devpack.inl) write the region record and blob of such a region
 * on demand, in the wave that is about to solve it (AvkKernelArgs::lazy_dp) */
__global__ void __launch_bounds__(256, AVK_LDS_WAVES_PER_SIMD) avk_region_kernel_lds_lazy(AvkKernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    const unsigned wave_in_block = threadIdx.x >> 6;
    const unsigned wave_id = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    if (a.esc_bytes) {
        for (unsigned k = threadIdx.x; k < AVK_WG_TAIL_BYTES / 4; k += blockDim.x) ((unsigned *)(avk_smem + a.esc_bytes))[k] = 0;
        __syncthreads();
    }
    avk::region_worker<true, true>(a, wave_id, avk_smem + (size_t)wave_in_block * a.tier[a.pass_tier].ws_bytes);
}
__global__ void __launch_bounds__(256, 2) avk_region_kernel_hbm_lazy(AvkKernelArgs a) {
    const unsigned wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    avk::region_worker<false, true>(a, wave_id, (unsigned char *)0);
}

/* Small regions, one per LANE (avk_lane.inl): a workgroup is four independent waves, each claims tiles of 64 fast records; the
 * workgroup's LDS holds the four waves' per-lane arrays and one shared tally that is flushed once */
#ifndef AVK_LANE_WPE
#define AVK_LANE_WPE 3
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AVK_LANE_WPE))) avk_lane_kernel(AvkKernelArgs a, avk::lane::LaneArgs la) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    uint32_t *smem = (uint32_t *)avk_smem;
    const unsigned wave_in_block = threadIdx.x >> 6;
    const unsigned wave_id = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const uint32_t rows = avk::lane::lane_rows(la.W, la.nm, la.ed_max, la.qcap, la.pool);
    const uint32_t wave_words = rows << la.lanes_log2;
    uint32_t *wg_tally = smem + (size_t)(blockDim.x >> 6) * wave_words;
    for (unsigned k = threadIdx.x; k < 288; k += blockDim.x) wg_tally[k] = 0;
    __syncthreads();
    uint32_t n_ok = 0, n_err = 0;
    uint64_t *part = a.tally + (uint64_t)(blockIdx.x % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE;
    avk::lane::lane_worker(a, la, wave_id, smem + (size_t)wave_in_block * wave_words, wg_tally, n_ok, n_err, blockDim.x == 64 ? part : (uint64_t *)0);
    n_ok = wv_sum_u32(n_ok);
    n_err = wv_sum_u32(n_err);
    if ((threadIdx.x & 63u) == 0) {
        if (n_ok) {
            atomicAdd((unsigned long long *)(part + AVK_TALLY_SOLVED), (unsigned long long)n_ok);
            atomicAdd((unsigned long long *)(part + AVK_TALLY_LANE_SOLVED), (unsigned long long)n_ok);
        }
        if (n_err) {
            atomicAdd((unsigned long long *)(part + AVK_TALLY_ERRORS), (unsigned long long)n_err);
            atomicAdd((unsigned long long *)(part + AVK_TALLY_LANE_SOLVED), (unsigned long long)n_err);
        }
    }
    __syncthreads();
    for (unsigned k = threadIdx.x; k < AVK_N_GROUPS * AVK_N_FIELDS; k += blockDim.x) {
        const uint32_t v = wg_tally[k];
        if (v) atomicAdd((unsigned long long *)(part + k), (unsigned long long)v);
    }
}

/* The expensive small regions — the heads of the lane classes and the three-call class — one per QUAD (avk_quad.inl): the same records, rows and
 * results as avk_lane_kernel at 16 (8, 4) records per wave, four lanes on every region instead of one */
#ifndef AVK_QUAD_WPE
#define AVK_QUAD_WPE 3
#endif
static __device__ __forceinline__ void avk_quad_body(const AvkKernelArgs &a, const avk::lane::LaneArgs &la) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    uint32_t *smem = (uint32_t *)avk_smem;
    const uint32_t rows = avk::quad::quad_rows(la.W, la.nm, la.ed_max, la.qcap, la.pool);
    const uint32_t wave_words = rows << la.lanes_log2;
    uint32_t *wg_tally = smem + wave_words;
    for (unsigned k = threadIdx.x; k < 288; k += blockDim.x) wg_tally[k] = 0;
    __syncthreads();
    uint32_t n_ok = 0, n_err = 0;
    uint64_t *part = a.tally + (uint64_t)(blockIdx.x % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE;
    avk::quad::quad_worker(a, la, blockIdx.x, smem, wg_tally, n_ok, n_err, part);
    n_ok = wv_sum_u32(n_ok);
    n_err = wv_sum_u32(n_err);
    if ((threadIdx.x & 63u) == 0) {
        if (n_ok) {
            atomicAdd((unsigned long long *)(part + AVK_TALLY_SOLVED), (unsigned long long)n_ok);
            atomicAdd((unsigned long long *)(part + AVK_TALLY_LANE_SOLVED), (unsigned long long)n_ok);
        }
        if (n_err) {
            atomicAdd((unsigned long long *)(part + AVK_TALLY_ERRORS), (unsigned long long)n_err);
            atomicAdd((unsigned long long *)(part + AVK_TALLY_LANE_SOLVED), (unsigned long long)n_err);
        }
    }
    __syncthreads();
    for (unsigned k = threadIdx.x; k < AVK_N_GROUPS * AVK_N_FIELDS; k += blockDim.x) {
        const uint32_t v = wg_tally[k];
        if (v) atomicAdd((unsigned long long *)(part + k), (unsigned long long)v);
    }
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AVK_QUAD_WPE))) avk_quad_kernel(AvkKernelArgs a, avk::lane::LaneArgs la) { avk_quad_body(a, la); }
/* The same code for launches whose LDS slice holds a CU to two waves per SIMD anyway (the three-call class: 20 KB per one-wave workgroup, seven per CU): built for two
 * waves per SIMD it has 256 registers and spills nothing (at three: 168 registers, 26 VGPR + 214 SGPR spills, 108 bytes of scratch per lane) */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) avk_quad_kernel_wide_regs(AvkKernelArgs a, avk::lane::LaneArgs la) { avk_quad_body(a, la); }

/* Regions with large searches on small windows, one per wave, every lane a piece of the search (avk_wide.inl): one-wave workgroups, the region's tables in
 * the workgroup's LDS */
__global__ void __launch_bounds__(64) avk_wide_kernel(AvkKernelArgs a, avk::wide::WideArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    avk::wide::wide_worker<false>(a, wa, blockIdx.x, (uint32_t *)avk_smem);
}
/* the same for regions the lanes handed back (device-packed batches write their records on demand, AvkKernelArgs::lazy_dp) */
__global__ void __launch_bounds__(64) avk_wide_kernel_lazy(AvkKernelArgs a, avk::wide::WideArgs wa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    avk::wide::wide_worker<true>(a, wa, blockIdx.x, (uint32_t *)avk_smem);
}

/* regions with the same SNV on both sides (avk_pairs.inl): table rows copied out, 64 regions per wave at a time */
__global__ void __launch_bounds__(256) avk_pair_kernel(AvkKernelArgs a, avk::pairs::PairArgs pa) {
    avk::pairs::pair_worker(a, pa, a.tally + (uint64_t)(blockIdx.x % AVK_TALLY_COPIES) * AVK_TALLY_STRIDE);
}

/* DWFALite scripts (avk_dwfa_script.inl): engine 0 one script per wavefront, engine 1 one script per lane */
__global__ void __launch_bounds__(256) avk_dwfa_wave_kernel(AvkDwfaArgs a) {
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave < a.n_scripts) avk::dwfa_script_wave(a, wave);
}
__global__ void __launch_bounds__(64) avk_dwfa_lane_kernel(AvkDwfaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char avk_smem[];
    const unsigned s = blockIdx.x * 64u + threadIdx.x;
    if (s < a.n_scripts) avk::lane::dwfa_script_lane(a, s, (uint32_t *)avk_smem);
}

/* Stratified tallies (SummaryWriter::add_comparison_benchmark with containment regions, writers/summary.rs:146-163): label l sums the
 * metric blocks of the solved regions whose label list names it.  One wave per region at a time; a workgroup keeps the tallies of the
 * AVK_LABEL_BLOCK labels of this launch in LDS (64-bit adds) and flushes them once — HBM traffic is the 1144-byte metric block of
 * every region that has a label in the block, read once. */
#define AVK_LABEL_BLOCK 16
#define AVK_GM_WORDS (AVK_N_GROUPS * AVK_N_FIELDS)
__global__ void __launch_bounds__(256) avk_label_tally_kernel(const uint32_t *gm, const uint32_t *region_out, const unsigned long long *label_off,
                                                             const uint32_t *label_idx, uint32_t n_regions, uint32_t label_lo, uint32_t label_hi,
                                                             unsigned long long *out) {
    __shared__ unsigned long long acc[AVK_LABEL_BLOCK * AVK_GM_WORDS];
    for (unsigned k = threadIdx.x; k < AVK_LABEL_BLOCK * AVK_GM_WORDS; k += blockDim.x) acc[k] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    for (unsigned r = wave; r < n_regions; r += n_waves) {
        if (region_out[4u * r] != 0) continue; /* only solved regions count */
        const unsigned long long lo = label_off[r], hi = label_off[r + 1];
        bool any = false;
        for (unsigned long long q = lo; q < hi; ++q) {
            const uint32_t l = label_idx[q];
            any = any || (l >= label_lo && l < label_hi);
        }
        if (!any) continue;
        uint32_t v[5];
        const uint32_t *block = gm + (size_t)r * AVK_GM_WORDS;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned i = lane + 64u * (unsigned)j;
            v[j] = i < AVK_GM_WORDS ? block[i] : 0u;
        }
        for (unsigned long long q = lo; q < hi; ++q) {
            const uint32_t l = label_idx[q];
            if (l < label_lo || l >= label_hi) continue;
            unsigned long long *dst = acc + (size_t)(l - label_lo) * AVK_GM_WORDS;
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (v[j]) atomicAdd(dst + lane + 64u * (unsigned)j, (unsigned long long)v[j]);
        }
    }
    __syncthreads();
    for (unsigned k = threadIdx.x; k < (label_hi - label_lo) * AVK_GM_WORDS; k += blockDim.x) {
        const unsigned long long x = acc[k];
        if (x) atomicAdd(out + (size_t)(label_lo + k / AVK_GM_WORDS) * AVK_TALLY_LEN + k % AVK_GM_WORDS, x);
    }
}

/* packs the uploaded reference: 16 bases per word, 2 bits each, plus one flag per word for anything that is
 * not an upper-case A/C/G/T (those windows are read from the byte copy).  One thread per packed word. */
__global__ void avk_pack_reference(const uint8_t *bytes, uint64_t n_bases, uint32_t *packed, uint32_t *exc) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_words = (n_bases + 15) >> 4;
    uint32_t word = 0, bad = 0;
    if (w < n_words) {
        for (int j = 0; j < 16; ++j) {
            const uint64_t p = w * 16 + j;
            uint32_t code = 0;
            if (p < n_bases) {
                const uint8_t ch = bytes[p];
                if (ch == 'A') code = 0;
                else if (ch == 'C') code = 1;
                else if (ch == 'G') code = 2;
                else if (ch == 'T') code = 3;
                else bad = 1;
            }
            word |= code << (2 * j);
        }
        packed[w] = word;
    }
    const unsigned long long m = __ballot(bad != 0);
    const unsigned lane = threadIdx.x & 63u;
    if (w < n_words + 64 && (lane & 31u) == 0) { /* 64 consecutive words = two 32-bit flag words */
        const uint64_t fw = w >> 5;
        if (fw <= ((n_words + 31) >> 5)) exc[fw] = lane == 0 ? (uint32_t)m : (uint32_t)(m >> 32);
    }
}

/* sums the partial tallies into out[0 .. AVK_TALLY_STRIDE) (and the caller's device tally, if any), then clears
 * the partial tallies and the work / overflow counters for the next call on this batch */
/* The records of class C (work order 0 .. n_c - 1) that are not for avk_wide.inl by what they say themselves (avk_wide_static_ok), as a list: a few dozen among
 * thousands, each a long search on a wave of the HBM tier.  Their launch used to walk the whole class in claims of four and solved the not-wide records of a claim one
 * after the other. */
__global__ void __launch_bounds__(256) avk_notwide_list_kernel(const AvkDevRegion *regions, uint32_t n_c, uint32_t *list, uint32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_c) return;
    const AvkDevRegion *g = regions + i;
    if (avk_wide_static_ok(g->len, g->grow, g->ed_bound, g->t_cnt, g->q_cnt, g->pre_status)) return;
    list[atomicAdd(count, 1u)] = i;
}

__global__ void avk_tally_reduce(uint64_t *partials, uint64_t *out, uint64_t *out_user, uint32_t *counters, unsigned n_counters, unsigned accumulate) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned k = i; k < n_counters; k += gridDim.x * blockDim.x) counters[k] = 0;
    if (i >= AVK_TALLY_STRIDE) return;
    uint64_t s = 0;
    for (int c = 0; c < AVK_TALLY_COPIES; ++c) {
        s += partials[(size_t)c * AVK_TALLY_STRIDE + i];
        partials[(size_t)c * AVK_TALLY_STRIDE + i] = 0;
    }
    out[i] = s;
    if (out_user && i < AVK_TALLY_LEN) out_user[i] = accumulate ? out_user[i] + s : s; /* accumulate: a job's running total over its batches */
}

/* ---------------------------------------------------------------------------------- context */
namespace {

std::string g_create_error;
std::mutex g_create_mutex;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct PoolBlk { /* a device buffer of the context's pool (avk_devpack_host.inl) */
    void *p;
    size_t bytes;
    bool used;
};

} // namespace

#define AVK_N_COUNTERS 1408
#define AVK_STREAM_ORDER_DEFAULT "stwxxabxxcd" /* profiles/r06_stream_order.txt */ /* words of avk_dev_batch::d_counters */

struct avk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    /* reference */
    uint8_t *d_ref = nullptr;
    uint32_t *d_ref2b = nullptr, *d_refexc = nullptr;
    int64_t use_packed_reference = 1;
    std::vector<uint64_t> contig_base, contig_len;
    uint64_t *d_contig_tab = nullptr; /* contig_base[n_contigs] then contig_len[n_contigs], for the device-side packer */
    /* device-side packing (avk_devpack_host.inl) */
    int64_t adaptive_ws = 1;               /* device-packed batches size their per-wave HBM slices by the predicted need of their class C regions (option ws_bytes_per_wave is the minimum) */
    int64_t ws_budget_bytes = 96ll << 30;  /* at most this much HBM for those slices: fewer waves per launch when the slices are large */
    int64_t device_pack = 1;               /* batches are validated, classified, ordered and written ON THE DEVICE from the caller's arrays (0: avk_pack.h on the host threads) */
    int64_t pool_cache_bytes = 12ll << 30; /* released device buffers the context keeps for the next batch (beyond: back to the runtime) */
    std::vector<PoolBlk> pool;
    std::mutex pool_mutex;
    size_t pool_free_bytes = 0;
    uint8_t *h_bounce = nullptr; /* pinned staging for caller arrays that are not pinned (grow-only) */
    size_t bounce_bytes = 0;
    void *h_dpstate = nullptr;   /* pinned: the packer's state block / the tally of a download */
    /* the asynchronous boundary (avk_compare_packed_submit / avk_wait): a copy stream each way and a ring of staging slots OUTSIDE the stream-ordered pool — the
     * packed arrays of batch k + 1 cross the bus while batch k is being solved, the results of batch k while batch k + 1 is being packed */
    hipStream_t copy_in_stream = nullptr, copy_out_stream = nullptr;
    hipStream_t pack_stream = nullptr, pack_side_stream = nullptr; /* the packing kernels of a submitted batch, beside the solver launches of the batch before */
    hipStream_t up_stream = nullptr, up_side = nullptr;            /* set while avk_compare_packed_submit runs upload_device_packed */
    hipEvent_t ev_packed = nullptr, ev_pool_fence = nullptr;
    bool pool_fence_pending = false; /* buffers went back to the pool in the order of the context's stream since the packing stream last waited for it */
    int64_t async_pack_stream = 1;                                 /* option: 0 = a submitted batch is packed on the context's stream, behind the solve of the batch before */
    struct StageSlot {
        uint8_t *dev = nullptr; /* one device block for the eleven arrays of an avk_packed_batch */
        size_t bytes = 0;
        uint64_t *h_tally = nullptr; /* pinned: the batch's tally lands here */
        hipEvent_t ev_in = nullptr, ev_unpacked = nullptr, ev_done = nullptr;
        bool busy = false;
    } stage[4];
    /* options */
    int64_t lds_bytes_per_wave = 10 * 1024;
    int64_t lds_ed_cap = 48;
    int64_t lds2_bytes_per_wave = 40 * 1024;
    int64_t lds2_ed_cap = 48;
    int64_t waves_per_cu = 16;
    int64_t solo_min_variants = 5; /* regions with at least this many variants go to solo waves (0 = no solo waves) */
    int64_t solo_blocks_max = 128;
    int64_t timing_events = 1; /* record the events avk_last_kernel_ms / avk_last_solver_ms read (three per call) */
    bool lds_attr_set = false;
    int64_t static_pct = AVK_STATIC_PCT; /* share of a launch's work list dealt statically; the rest is claimed */
    int64_t claim = AVK_CLAIM;           /* regions per claim */
    int64_t class_c_below = 16384; /* a batch with lane launches and at most this many regions outside them plans those regions as class C: the wide kernel (0: no such rule) */
    int64_t class_c_nodes_x2 = 12; /* a region is sent to the HBM solo launch when 0.5 x this x N nodes outgrow a tier-1 slice */
    int64_t solo_regions_per_wave = 4; /* predicted-hard regions beyond solo waves x this lead the bulk list */
    int64_t accumulate_tally = 0; /* avk_compare_resident adds to the caller's device tally instead of overwriting it */
    int64_t lds_escalation = 1; /* in-workgroup escalation of the bulk launch (AvkKernelArgs::esc_bytes) */
    int64_t lds2_overflow_pass = 0; /* 1: a launch of its own with large LDS slices between the bulk and the HBM tier */
    int64_t bulk_full_grid = 0; /* 1: keep the bulk grid at full size (late workgroups only claim); measured unstable */
    int64_t bulk_fit = 1;       /* 1: no more workgroups in the bulk launch than its list has regions for (a genome leaves it three dozen regions: 664 workgroups of 40 KB of
                                   LDS queued for them beside the lane launches) */
    int64_t ws_bytes_per_wave = 1 << 20;
    int64_t big_ws_bytes = 64ll << 20; /* (256 MB until round 5: the shared slices were 2.1 GB of a fresh process's first hipMalloc; a batch whose packer predicts larger regions gets 1 GB slices, upload_device_packed) */
    int64_t big_waves = 8;
    int64_t emit_group_metrics = 1;
    int64_t team_long_windows = 1;         /* 1: the launch of the long windows runs a workgroup per region (avk_region_kernel_team); 0: a wave per region (round 5) */
    int64_t team_head_regions = 48;        /* a batch of large windows (no region of class C is the wide kernel's): this many regions at the head of the class go to a team launch */
    int64_t split_parts = 1;               /* > 1: a large avk_compare_packed call runs as this many batches in flight (compare_packed_split; measured slower than the whole call while a half genome's step costs 2.0 of the whole's 2.4 ms: profiles/r06_split_call.txt) ... */
                                           /* ... when every part has at least 4096 regions and every array of the caller's is pinned */
    int64_t copy_blocks_per_cu = 8; /* workgroups of avk_copy_kernel per CU (AVK_COPY_BLOCKS in the environment overrides: a tuning aid) */
    int64_t kernel_copies = 1;  /* the pinned arrays of a synchronous call cross the bus 0: by the DMA engine the process drew, 2: by a copy kernel, 1: by whichever a measurement of
                                   the engine says (avk_devpack_host.inl: copies_by_kernel) */
    bool engines_fast = true;   /* no call of this context has seen its arrays cross below 36 GB/s on the engine */
    double engine_in_gbs = 0;
    size_t cp_timed_bytes = 0;  /* bytes between ev_cp0 and ev_cp1 of the last engine-timed copy_in (0: none pending) */
    hipEvent_t ev_cp0 = nullptr, ev_cp1 = nullptr;
    int64_t packed_source = 1;  /* 1: a batch in the packed form is packed from the packed arrays themselves (no wide copy of the caller's arrays in HBM); 0: round 5's widening pass */
    int64_t emit_bp_groups = 0; /* kernels write the compact per-region BASEPAIR groups (avk_result_batch::bp_groups) */
    int64_t capacity_retry = 1; /* avk_results_download solves regions that exhausted the last workspace tier again with larger slices */
    int64_t lane_kernel = 1; /* small regions go to the lane-per-region kernel (avk_lane.inl) */
    int64_t lane_min_regions = 2048; /* a lane class is launched when it holds at least this many regions (x16 for the two-call classes, x2 for the three-call
                                        class): a launch lasts at least as long as its slowest tile, which a small class cannot amortise.  8192 until the end of round 4;
                                        since the lanes keep node states and skip mirror-image optima their classes pay at a quarter of the size: an eighth-of-a-genome
                                        step 2.0 -> 1.5 ms, a quarter-genome step 3.6 -> 2.15 ms, the whole genome unchanged (profiles/r04_lane_min.txt) */
    int64_t lane_width_one = 64, lane_width_two = 64, lane_width_three = 16; /* records a wave takes at a time (64, 32, 16) in the one- / two- / three-call classes */
    int64_t lane_max_calls = AVK_FAST_MAXV;           /* classes with more calls per side stay with the wave-per-region kernels */
    int64_t lane_max_est = 15;                        /* regions whose estimated edits (fast_cost_key, avk_pack.h) exceed this stay with the wave-per-region kernels */
    int64_t pair_classes = 1;                         /* 1: pair batches (merge) plan their large searches as classes C / B like compare batches do, so that the wide kernel
                                                         and the solo launches take them at the start of the step; 0 (until the end of round 4): everything the lanes do not take
                                                         goes through the bulk and what overflows there through an HBM launch at the very end — a 3-caller whole-genome merge
                                                         call 22.5 -> 16 ms */
    int64_t lane_head_auto = 1;                       /* 1: a head of fewer regions than the machine has lane waves for takes fewer records per wave (lane_head_width is the most):
                                                         under 24,576 regions 8, under 8,192 regions 4 — an eighth-of-a-genome step 1.5 -> 1.2 ms, a quarter genome 2.15 -> 1.95 */
    int64_t lane_head_width = 16;                     /* records a wave takes at a time in the HEAD of a lane class: the tiles of regions with estimated edits (0 = no head launch) */
    int64_t lane_metrics_ed_cap = 0;                  /* lanes hand a region over when an alignment of its metrics phase passes this distance (0 = as far as the LDS rows allow: 30 / 54) */
    int64_t hbm_ed_cap = 1024;                        /* the per-wave HBM slices size a region's wavefronts by the region's own bound (the sum of its calls' edit distances) when that is at most this;
                                                         0: two entries per base of the window for every region (rounds 1-2).  Large windows: 1.64 -> 1.21 s per 49 k-region step, whole genome -3 % */
    int64_t het_search_min = AVK_HET_SEARCH_MIN;      /* regions with at least this many unphased heterozygous calls go to class C and stay out of the three-call lane class (0 = no such rule) */
    int64_t lane_head_est = 1;                        /* regions with at least this many estimated edits (fast_cost_key) form the narrow-tiled head of their lane class */
    int64_t lane_pairs = 1;                           /* regions with the same SNV on both sides are looked up in a table the solver fills (avk_pairs.inl); 0: they stay in the one-call classes */
    avk::pairs::PairTable *d_pair_tab = nullptr;      /* the table, made for max_branch_factor pair_tab_mbf */
    uint32_t *d_pair_aux = nullptr;                   /* probe records, probe reference and the scratch outputs of the probe launch */
    int64_t pair_tab_mbf = -1;
    int64_t pair_blocks_per_cu = 4;                   /* workgroups (4 waves) of the lookup launch per CU */
    int64_t lane_stripe = 0;                          /* 1: the heads' records dealt out over their claims (avk_stripe_slot, avk_dev_types.h) instead of most expensive first.  Measured
                                                         WORSE (whole genome 4.84 -> 5.40 ms per step): regions of one cost key take the same path through the search, a claim of equals
                                                         runs in lockstep, a claim of unequals takes turns */
    int64_t lane_head_stream = 0;                     /* 1: the heads of the two-call classes on a stream of their own (a synchronised step: 5.4 -> 5.1 ms;
                                                         steps queued back to back: 6.0 -> 6.5 ms — more streams, worse starts; off) */
    int64_t lane_min_batch = 16384;                   /* a RESIDENT batch with fewer lane regions than this is solved by the wave-per-region kernels alone (not applied when lane_min_regions is 0, nor by the one-shot path of avk_compare_batch) */
    int64_t hbm_early_blocks = 64;                    /* workgroups (x 4 waves, 1 MB of HBM workspace each) of the launch behind the three-call lane class */
    int64_t hbm_solo_blocks = 64;                     /* most workgroups (x 4 waves, 1 MB of HBM workspace each) of the HBM solo launch */
    int64_t lane_node_cap = 32;                       /* search nodes the three-call lane class makes before it hands a region over */
    int64_t lane_quad = 1;                            /* 1: lane launches of at most 16 records per wave (the heads, the three-call class) run four lanes per region: avk_quad_kernel, avk_quad.inl */
    int64_t lane_pool = -1;                           /* node states a lane keeps during its search (avk_lane.inl NodePool): -1 = by class (2 / 4 / 6 for one / two / three calls per side), 0 = none */
    int64_t lane_waves_three = 0;                     /* > 0: at most this many one-wave workgroups of the three-call class per CU (its waves take 17 KB of LDS each) */
    int64_t lane_waves_per_cu = 12;                   /* at most this many one-wave workgroups of a lane launch per CU */
    int64_t wide_kernel = 1;                          /* regions with large searches on small windows (class C, what the three-call lane class hands back) go to the wave-cooperative kernel of avk_wide.inl first */
    int64_t wide_lds_bytes = 16 * 1024;               /* LDS of one of its waves: 1.2 KB of tables, the region's 2^T + 2^Q full-length sequences, the rest search nodes (10 words each, at most 240) and 32 wavefront blocks */
    int64_t wide_retry_lds_bytes = 64 * 1024;         /* LDS per wave of the second launch over what the class C launch handed over (0: none) */
    int64_t wide_blocks = 512;                        /* most one-wave workgroups of its class C launch */
    int64_t wide_lane_handbacks = 1;                  /* what the one- and two-call lane classes hand back goes through avk_wide.inl too, ahead of the LDS launch */
    int64_t wide_lazy_blocks = 512;                   /* one-wave workgroups of its launch for what the three-call lane class hands back (a short list, its length known on the device only) */
    bool wide_attr_set = false;
    uint64_t last_wide_solved = 0;
    bool lane_attr_set = false;
    uint64_t last_lane_solved = 0;
    int last_one_shot = 0; /* the last avk_compare_batch / avk_optimize_pairs_batch packed its batch on the device */
    int n_cus = 0;
    /* workspaces (grown on demand) */
    uint8_t *d_ws = nullptr;
    size_t ws_alloc = 0;
    uint8_t *d_big = nullptr;
    size_t big_alloc = 0;
    /* measurement */
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk1 = nullptr; /* ev0..ev1 all solver launches, ev0..evk1 the first (dominant) one */
    hipEvent_t ev_lane = nullptr; /* after the lane-kernel launches of a step */
    bool ev_lane_valid = false;
    hipStream_t lane_stream = nullptr, lane_stream2 = nullptr; /* the lane-kernel launches run beside the wave-per-region launches: two-call classes / one-call classes */
    hipEvent_t ev_lane_fork = nullptr, ev_lane_join = nullptr, ev_lane_join2 = nullptr;
    hipStream_t lane_stream3 = nullptr; /* the three-call class: long tiles, few of them, beside everything else */
    hipEvent_t ev_lane_join3 = nullptr, ev_lane_done = nullptr;
    hipStream_t lane_stream4 = nullptr; /* the head launches of the two-call classes (long: beside the rest of their class, not ahead of it) */
    hipEvent_t ev_lane_join4 = nullptr, ev_lane_early = nullptr;
    hipEvent_t ev_hb[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; /* ends of the lane classes' head launches (handback_chains) */
    hipEvent_t ev_tl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; /* AVK_TIMING only: marks of a boundary call on the context's stream (first copy, last copy, work order, writers, results) */
    hipEvent_t ev_copy_alleles = nullptr; /* packed upload: behind the allele bytes' copy (dp_variant starts there) */
    hipEvent_t ev_copy_fork = nullptr, ev_copy_mid = nullptr, ev_copy_join = nullptr; /* packed upload: all but the counts cross on lane_stream4 beside the offset kernels */
    hipStream_t side_stream = nullptr, side_stream2 = nullptr; /* solo launches (LDS, HBM): one stream each, they run side by side */
    hipStream_t spare_stream[16] = {nullptr}; /* never used: see avk_ctx_create */
    int n_placeholders = 0;                   /* how many of them avk_ctx_create made (and destroyed again) */
    hipStream_t wide_stream = nullptr; /* the class C records that are not for the wide kernel (run_internal) */
    std::thread reaper; /* releases the buffers of the last large batch behind the caller (avk_batch_free) */
    std::mutex reaper_mutex;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_wide = nullptr;
    bool ev_valid = false;
    uint64_t last_tiers[5] = {0, 0, 0, 0, 0};
    uint64_t last_phase[16] = {0};
};

struct avk_dev_batch {
    avk::PackedBatch host; /* regions/variants kept for the scatter maps */
    uint64_t n_regions = 0, n_variants_dev = 0, n_variants_host = 0;
    bool want_seq = false;
    uint64_t seq_total = 0;
    AvkDevRegion *d_regions = nullptr;
    uint32_t *d_blob = nullptr; /* region blobs (AvkBlobVar in avk_dev_types.h) */
    uint32_t *d_region_out = nullptr; /* [n][4] */
    uint32_t *d_gm = nullptr;
    uint32_t *d_var_out = nullptr;    /* [nv] */
    uint8_t *d_seq = nullptr;
    uint32_t *d_seqlen = nullptr;
    uint64_t *d_tally = nullptr;    /* [AVK_TALLY_STRIDE]: AVK_TALLY_LEN sums, 5 tier counters, 8 profiling words */
    uint64_t *d_partials = nullptr; /* [AVK_TALLY_COPIES][AVK_TALLY_STRIDE] */
    uint32_t *d_counters = nullptr; /* [256*t + 32*s] claim counter of shard s in pass t, [1024 + 16*k] overflow counts, [1072] claim counter of the solo waves */
    uint32_t *d_overflow = nullptr, *d_overflow2 = nullptr, *d_overflow3 = nullptr, *d_overflow4 = nullptr;
    uint32_t *d_notwide = nullptr;   /* [n + 1] class C records that are not avk_wide.inl's, then their number (avk_notwide_list_kernel, once per batch) */
    bool notwide_ready = false;
    uint32_t *d_overflow8 = nullptr; /* what the second, large-LDS launch of avk_wide.inl over class C's leftovers could not take either */
    uint32_t *d_overflow5 = nullptr, *d_overflow6 = nullptr, *d_overflow7 = nullptr; /* what the launches of avk_wide.inl could not take: of class C, of the three-call lane class's hand-backs, of the other lane classes' */
    avk::WorkPlan plan;
    uint32_t *d_fast = nullptr; /* fast records of the lane-per-region kernel (avk_dev_types.h), tiles of 64 */
    uint64_t fast_word_base[AVK_FAST_CLASSES] = {0}; /* first word of the class's tiles in d_fast */
    uint32_t fast_tiles[AVK_FAST_CLASSES] = {0};
    avk_compare_config last_cfg = {50, 0, 0}; /* the configuration of the last run (the capacity retry of avk_results_download repeats it) */
    uint32_t last_mode = 0;
    bool has_run = false;
    bool scratch_clean = false; /* partial tallies and counters are zero */
    bool with_gm = true;
    /* device-packed batches (avk_devpack_host.inl): every buffer comes from the context's pool; `host` stays empty unless the capacity retry or
     * the sequence outputs ask for it (materialize_host_view) */
    bool dev_packed = false;
    std::vector<void *> pooled;
    uint64_t *d_in_t_off = nullptr, *d_in_q_off = nullptr; /* the caller's t_off / q_off / t_cnt / q_cnt: dp_unpack scatters the per-call outputs with them */
    uint32_t *d_in_t_cnt = nullptr, *d_in_q_cnt = nullptr, *d_voff = nullptr;
    const uint64_t *d_pk_voff = nullptr; /* a compare batch read from its packed source (DpIn::pk_*): these three stand in for the four arrays above */
    const uint8_t *d_pk_tc = nullptr, *d_pk_qc = nullptr;
    avk::dp::DpArgs dp_args; /* the packer's arguments: the writers of region records run again for the regions a launch turns out to need */
    avk::dp::DpArgs *d_dp_args = nullptr; /* the same in device memory (AvkKernelArgs::lazy_dp) */
    uint32_t lazy_from = 0;               /* first work-order index without a record (the lane classes' segment) */
    uint32_t *d_bp_off = nullptr, *d_bp = nullptr; /* compact per-region BASEPAIR groups: first group of a region (caller order), the groups (allocated when a run asks for them) */
    uint64_t n_bp_groups = 0;
    uint64_t *d_m_in_off = nullptr;       /* merge batches (avk_merge_batch): the MultiRegions' in_off / in_cnt and the calls' zygosities, for the classification kernel */
    uint32_t *d_m_in_cnt = nullptr;
    uint8_t *d_in_zyg = nullptr;
    uint64_t n_multi = 0;
    uint32_t m_inputs = 0;
    uint64_t v_lo = 0, v_hi = 0;          /* the calls the batch's regions own: results are copied back for this range of the caller's arrays only */
    bool var_dense = false;               /* every call of the range is owned by a region of the batch */
    int64_t big_bytes_eff = 0;            /* shared big slices of this batch's launches when the packer predicts regions beyond the largest bucket (0: the option big_ws_bytes) */
    int64_t ws_bytes_eff = 0;             /* per-wave HBM slice of this batch's launches when the packer's prediction asks for more than the option ws_bytes_per_wave (0: the option) */
    bool records_full = false; /* every region has its AvkDevRegion + blob (false: only the regions outside the lane classes, until a launch asks for more) */
};

namespace {

int fail(avk_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else {
        std::lock_guard<std::mutex> lk(g_create_mutex);
        g_create_error = buf;
    }
    return code;
}

#define AVK_HIP(ctx, call)                                                                                  \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? AVK_E_OOM : AVK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T> int dev_alloc(avk_ctx *ctx, T **p, size_t count) {
    *p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    AVK_HIP(ctx, hipMalloc((void **)p, bytes));
    return 0;
}

void free_batch_buffers(avk_dev_batch *db) {
    if (db->dev_packed) return; /* pooled buffers: release_pooled */
    void *ptrs[] = {db->d_regions, db->d_blob, db->d_region_out, db->d_gm, db->d_var_out,
                    db->d_seq, db->d_seqlen, db->d_tally, db->d_partials, db->d_counters, db->d_overflow, db->d_overflow2, db->d_overflow3, db->d_overflow4, db->d_overflow5, db->d_overflow6, db->d_overflow7, db->d_overflow8, db->d_notwide, db->d_fast,
                    db->d_bp_off, db->d_bp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

} // namespace

#include "avk_devpack_host.inl"

extern "C" {

const char *avk_version(void) { return "aardvark_amd 0.1 (gfx950)"; }
#ifndef AVK_SOURCE_HASH
#define AVK_SOURCE_HASH "unknown"
#endif
const char *avk_source_hash(void) { return AVK_SOURCE_HASH; }

uint64_t avk_edit_distance(const uint8_t *a, uint64_t a_len, const uint8_t *b, uint64_t b_len) { return avk::host_edit_distance(a, a_len, b, b_len); }

const char *avk_last_error(const avk_ctx *ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_create_mutex);
    return g_create_error.c_str();
}

int avk_ctx_create(int device_id, avk_ctx **out) {
    if (!out) return AVK_E_ARG;
    *out = nullptr;
    /* six streams side by side need hardware queues of their own (the runtime's default is 4 and streams that share one run in turn);
     * read by the runtime when it initialises, so this helps when this is the process's first HIP call; the caller's setting wins */
    setenv("GPU_MAX_HW_QUEUES", "24", 0); /* 8 are enough for this library alone; with a communicator (RCCL) in the process 8 make the step 6.0 ms instead of 3.8 */
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, AVK_E_HIP, "no HIP device available (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, AVK_E_ARG, "device %d out of range (have %d)", device_id, n);
    avk_ctx *ctx = new avk_ctx();
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, AVK_E_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, AVK_E_HIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    }
    ctx->n_cus = prop.multiProcessorCount;
    if (const char *e = getenv("AVK_COPY_BLOCKS")) ctx->copy_blocks_per_cu = atoi(e) > 0 ? atoi(e) : ctx->copy_blocks_per_cu;
    if (const char *e = getenv("AVK_WIDE_BLOCKS")) ctx->wide_blocks = atoi(e) > 0 ? atoi(e) : ctx->wide_blocks;
    if (hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return fail(nullptr, AVK_E_HIP, "hipStreamCreate failed");
    }
    ctx->own_stream = true;
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreate(&ctx->evk1) != hipSuccess ||
        hipEventCreate(&ctx->ev_lane) != hipSuccess ||
        /* the buffer pool's fence exists from the start: a resident upload that releases its temporaries in stream order BEFORE the first asynchronous submit must be
         * able to record it (upload_device_packed), or the first packing on pack_stream could be handed buffers that queued kernels still read */
        hipEventCreateWithFlags(&ctx->ev_pool_fence, hipEventDisableTiming) != hipSuccess) {
        avk_ctx_destroy(ctx);
        return fail(nullptr, AVK_E_HIP, "hipEventCreate failed");
    }
    /* the solo launches get streams of the highest priority: they are the critical path, and HIP never folds streams of
     * different priorities onto one hardware queue (with a communicator library in the process the default-priority streams
     * of a process share queues, and a shared queue would serialise the solo launches with the bulk) */
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    { /* the side and lane streams run at the default priority: with hardware queues of their own (GPU_MAX_HW_QUEUES >= 8) a high priority changes
       * nothing (4.94 / 4.98 ms per whole-genome step), with the runtime's default of 4 queues it costs 0.9 ms (7.0 / 7.9 ms);
       * AVK_STREAM_PRIORITY=high brings it back for experiments */
        const char *pe = getenv("AVK_STREAM_PRIORITY");
        if (!pe || strcmp(pe, "high") != 0) prio_high = 0;
    }
    const unsigned evf = getenv("AVK_TIMING") ? hipEventDefault : hipEventDisableTiming; /* the events that end the launch chains can be read when the stage timing is on */
    /* The runtime hands streams their hardware queues in the order they are made, and which launches share a queue's pipe shows in the step: streams that nothing is
     * ever queued on, made between the others, move a shard's resident step between 1.05 and 1.64 ms and a genome's between 2.38 and 2.95 (profiles/r05_stream_order.txt,
     * profiles/r06_stream_order.txt: a local search over the order, every candidate in fresh processes).  AVK_STREAM_ORDER: the order the seven side streams are made in,
     * a letter each — s side, t side2, w wide, a b c d the four lane streams, x a stream nothing is ever queued on; i o p q: the copy-in, copy-out, packing and
     * packing-side streams of the asynchronous calls (made at the first submit when the order does not name them). */
    const char *order = getenv("AVK_STREAM_ORDER") ? getenv("AVK_STREAM_ORDER") : AVK_STREAM_ORDER_DEFAULT;
    int n_spare = 0;
    bool ok = true;
    for (const char *o = order; *o && ok; ++o) {
        hipStream_t *st = *o == 's' ? &ctx->side_stream : *o == 't' ? &ctx->side_stream2 : *o == 'w' ? &ctx->wide_stream : *o == 'a' ? &ctx->lane_stream : *o == 'b' ? &ctx->lane_stream2 :
                          *o == 'c' ? &ctx->lane_stream3 : *o == 'd' ? &ctx->lane_stream4 : *o == 'i' ? &ctx->copy_in_stream : *o == 'o' ? &ctx->copy_out_stream :
                          *o == 'p' ? &ctx->pack_stream : *o == 'q' ? &ctx->pack_side_stream : (*o == 'x' && n_spare < 16) ? &ctx->spare_stream[n_spare++] : nullptr;
        if (st && !*st) ok = hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_high) == hipSuccess;
    }
    /* the placeholders have done their work once the others exist: the streams made behind them keep their place, and a process that held on to four idle streams
     * pushed a second process on the same GPU over the number of hardware queues the scheduler maps at a time — the tool's solve stage 0.02 -> 0.10-0.21 s beside
     * bench.py's own context (profiles/r06_stream_order.txt).  AVK_KEEP_SPARES=1 keeps them (to reproduce that). */
    ctx->n_placeholders = n_spare;
    if (!getenv("AVK_KEEP_SPARES"))
        for (int k = 0; k < n_spare; ++k) {
            if (ctx->spare_stream[k]) (void)hipStreamDestroy(ctx->spare_stream[k]);
            ctx->spare_stream[k] = nullptr;
        }
    hipStream_t *all[7] = {&ctx->side_stream, &ctx->side_stream2, &ctx->wide_stream, &ctx->lane_stream, &ctx->lane_stream2, &ctx->lane_stream3, &ctx->lane_stream4};
    for (hipStream_t *st : all) /* (a letter the order left out) */
        if (ok && !*st) ok = hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_high) == hipSuccess;
    if (!ok ||
        hipEventCreateWithFlags(&ctx->ev_join2, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_wide, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_join2, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_join3, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_done, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_join4, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_hb[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_hb[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_hb[2], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_hb[3], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_hb[4], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_hb[5], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_hb[6], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_hb[7], hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&ctx->ev_cp0) != hipSuccess || hipEventCreate(&ctx->ev_cp1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_copy_alleles, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_copy_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_copy_mid, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_copy_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_early, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lane_join, evf) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, evf) != hipSuccess) {
        avk_ctx_destroy(ctx);
        return fail(nullptr, AVK_E_HIP, "cannot create the side streams / events of the context");
    }
    *out = ctx;
    return 0;
}

void avk_ctx_destroy(avk_ctx *ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(ctx->reaper_mutex);
        if (ctx->reaper.joinable()) ctx->reaper.join();
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &sl : ctx->stage) { /* staging slots of the asynchronous boundary */
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.h_tally) (void)hipHostFree(sl.h_tally);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_unpacked) (void)hipEventDestroy(sl.ev_unpacked);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (ctx->pack_stream) (void)hipStreamDestroy(ctx->pack_stream);
    if (ctx->pack_side_stream) (void)hipStreamDestroy(ctx->pack_side_stream);
    if (ctx->ev_packed) (void)hipEventDestroy(ctx->ev_packed);
    if (ctx->ev_pool_fence) (void)hipEventDestroy(ctx->ev_pool_fence);
    if (ctx->copy_in_stream) (void)hipStreamDestroy(ctx->copy_in_stream);
    if (ctx->copy_out_stream) (void)hipStreamDestroy(ctx->copy_out_stream);
    pool_destroy(ctx);
    if (ctx->d_contig_tab) (void)hipFree(ctx->d_contig_tab);
    if (ctx->d_ref) (void)hipFree(ctx->d_ref);
    if (ctx->d_ref2b) (void)hipFree(ctx->d_ref2b);
    if (ctx->d_refexc) (void)hipFree(ctx->d_refexc);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_big) (void)hipFree(ctx->d_big);
    for (hipEvent_t &t : ctx->ev_tl)
        if (t) (void)hipEventDestroy(t);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evk1) (void)hipEventDestroy(ctx->evk1);
    if (ctx->ev_lane) (void)hipEventDestroy(ctx->ev_lane);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_wide) (void)hipEventDestroy(ctx->ev_wide);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    if (ctx->d_pair_tab) (void)hipFree(ctx->d_pair_tab);
    if (ctx->d_pair_aux) (void)hipFree(ctx->d_pair_aux);
    if (ctx->ev_lane_fork) (void)hipEventDestroy(ctx->ev_lane_fork);
    if (ctx->ev_lane_join) (void)hipEventDestroy(ctx->ev_lane_join);
    if (ctx->ev_lane_join2) (void)hipEventDestroy(ctx->ev_lane_join2);
    if (ctx->lane_stream) (void)hipStreamDestroy(ctx->lane_stream);
    if (ctx->lane_stream2) (void)hipStreamDestroy(ctx->lane_stream2);
    if (ctx->ev_lane_join3) (void)hipEventDestroy(ctx->ev_lane_join3);
    if (ctx->ev_lane_done) (void)hipEventDestroy(ctx->ev_lane_done);
    if (ctx->ev_lane_join4) (void)hipEventDestroy(ctx->ev_lane_join4);
    for (hipEvent_t e : ctx->ev_hb)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev_cp0) (void)hipEventDestroy(ctx->ev_cp0);
    if (ctx->ev_cp1) (void)hipEventDestroy(ctx->ev_cp1);
    if (ctx->ev_copy_alleles) (void)hipEventDestroy(ctx->ev_copy_alleles);
    if (ctx->ev_copy_fork) (void)hipEventDestroy(ctx->ev_copy_fork);
    if (ctx->ev_copy_mid) (void)hipEventDestroy(ctx->ev_copy_mid);
    if (ctx->ev_copy_join) (void)hipEventDestroy(ctx->ev_copy_join);
    if (ctx->ev_lane_early) (void)hipEventDestroy(ctx->ev_lane_early);
    if (ctx->lane_stream4) (void)hipStreamDestroy(ctx->lane_stream4);
    if (ctx->lane_stream3) (void)hipStreamDestroy(ctx->lane_stream3);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->side_stream2) (void)hipStreamDestroy(ctx->side_stream2);
    if (ctx->wide_stream) (void)hipStreamDestroy(ctx->wide_stream);
    for (hipStream_t sp : ctx->spare_stream)
        if (sp) (void)hipStreamDestroy(sp);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int avk_ctx_set_stream(avk_ctx *ctx, void *hip_stream) {
    if (!ctx) return AVK_E_ARG;
    (void)hipSetDevice(ctx->device);
    /* the buffer pool and the bounce buffer hand memory out again in the order of ONE stream: whatever the previous stream still has queued (the writers of an
     * upload's temporaries, say) is finished before another stream may be given the same buffers — also when the previous stream was the caller's */
    if (ctx->stream != (hipStream_t)hip_stream) {
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            /* a caller's stream that is gone already (the header asks for it to be alive; the handle must not stay installed either way): whatever was queued on the
             * device is finished instead, and the switch goes through */
            (void)hipGetLastError();
            if (ctx->own_stream) return fail(ctx, AVK_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
            AVK_HIP(ctx, hipDeviceSynchronize());
        }
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return 0;
}

int avk_ctx_set_option(avk_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return AVK_E_ARG;
    std::string n(name);
    if (n == "lds_bytes_per_wave") {
        if (value < 0 || value > 40 * 1024) return fail(ctx, AVK_E_ARG, "lds_bytes_per_wave must be in [0, 40960]");
        ctx->lds_bytes_per_wave = value & ~15ll;
    } else if (n == "lds2_bytes_per_wave") {
        if (value < 0 || value > 40 * 1024) return fail(ctx, AVK_E_ARG, "lds2_bytes_per_wave must be in [0, 40960]");
        ctx->lds2_bytes_per_wave = value & ~15ll;
    } else if (n == "lds2_ed_cap") {
        if (value < 1 || value > 4096) return fail(ctx, AVK_E_ARG, "lds2_ed_cap must be in [1, 4096]");
        ctx->lds2_ed_cap = value;
    } else if (n == "lds_ed_cap") {
        if (value < 1 || value > 4096) return fail(ctx, AVK_E_ARG, "lds_ed_cap must be in [1, 4096]");
        ctx->lds_ed_cap = value;
    } else if (n == "solo_min_variants") {
        if (value < 0) return fail(ctx, AVK_E_ARG, "solo_min_variants must not be negative");
        ctx->solo_min_variants = value;
    } else if (n == "accumulate_tally") {
        ctx->accumulate_tally = value ? 1 : 0;
    } else if (n == "lds_escalation") {
        ctx->lds_escalation = value ? 1 : 0;
    } else if (n == "class_c_below") {
        if (value < 0) return fail(ctx, AVK_E_ARG, "class_c_below must not be negative");
        ctx->class_c_below = value;
    } else if (n == "class_c_nodes_x2") {
        if (value < 1 || value > 1000) return fail(ctx, AVK_E_ARG, "class_c_nodes_x2 must be in [1, 1000]");
        ctx->class_c_nodes_x2 = value;
    } else if (n == "static_pct") {
        if (value < 0 || value > 100) return fail(ctx, AVK_E_ARG, "static_pct must be in [0, 100]");
        ctx->static_pct = value;
    } else if (n == "claim") {
        if (value < 1 || value > 64) return fail(ctx, AVK_E_ARG, "claim must be in [1, 64]");
        ctx->claim = value;
    } else if (n == "wide_kernel") {
        ctx->wide_kernel = value ? 1 : 0;
    } else if (n == "wide_lds_bytes") {
        if (value < 8 * 1024 || value > 64 * 1024) return fail(ctx, AVK_E_ARG, "wide_lds_bytes must be in [8192, 65536]");
        ctx->wide_lds_bytes = value & ~15ll;
    } else if (n == "wide_retry_lds_bytes") {
        if (value < 0 || value > 64 * 1024) return fail(ctx, AVK_E_ARG, "wide_retry_lds_bytes must be in [0, 65536]");
        ctx->wide_retry_lds_bytes = value & ~15ll;
    } else if (n == "waves_per_cu") {
        if (value < 1 || value > 32) return fail(ctx, AVK_E_ARG, "waves_per_cu must be in [1, 32]");
        ctx->waves_per_cu = value;
    } else if (n == "ws_bytes_per_wave") {
        if (value < 0) return fail(ctx, AVK_E_ARG, "ws_bytes_per_wave must be >= 0");
        ctx->ws_bytes_per_wave = (value + 255) & ~255ll;
    } else if (n == "big_ws_bytes") {
        if (value < 0) return fail(ctx, AVK_E_ARG, "big_ws_bytes must be >= 0");
        ctx->big_ws_bytes = (value + 255) & ~255ll;
    } else if (n == "use_packed_reference") {
        ctx->use_packed_reference = value ? 1 : 0;
    } else if (n == "adaptive_ws") {
        ctx->adaptive_ws = value ? 1 : 0;
    } else if (n == "ws_budget_bytes") {
        if (value < (1ll << 30)) return fail(ctx, AVK_E_ARG, "ws_budget_bytes must be at least 1 GiB");
        ctx->ws_budget_bytes = value;
    } else if (n == "device_pack") {
        ctx->device_pack = value ? 1 : 0;
    } else if (n == "team_long_windows") {
        if (value < 0 || value > 2) return fail(ctx, AVK_E_ARG, "team_long_windows must be 0, 1 or 2 (2: the owner wave takes every job itself, a diagnostic)");
        ctx->team_long_windows = value;
    } else if (n == "team_head_regions") {
        if (value < 0 || value > 1024) return fail(ctx, AVK_E_ARG, "team_head_regions must be 0..1024");
        ctx->team_head_regions = value;
    } else if (n == "split_parts") {
        if (value < 1 || value > 4) return fail(ctx, AVK_E_ARG, "split_parts must be 1..4");
        ctx->split_parts = value;
    } else if (n == "kernel_copies") {
        if (value < 0 || value > 2) return fail(ctx, AVK_E_ARG, "kernel_copies must be 0 (engines), 1 (by measurement) or 2 (kernel)");
        ctx->kernel_copies = value;
    } else if (n == "packed_source") {
        ctx->packed_source = value ? 1 : 0;
    } else if (n == "emit_bp_groups") {
        ctx->emit_bp_groups = value ? 1 : 0;
    } else if (n == "emit_group_metrics") {
        ctx->emit_group_metrics = value ? 1 : 0;
    } else if (n == "capacity_retry") {
        ctx->capacity_retry = value ? 1 : 0;
    } else if (n == "lane_kernel") {
        ctx->lane_kernel = value ? 1 : 0;
    } else if (n == "lane_min_regions") {
        if (value < 0) return fail(ctx, AVK_E_ARG, "lane_min_regions must not be negative");
        ctx->lane_min_regions = value;
    } else if (n == "lane_width_one" || n == "lane_width_two" || n == "lane_width_three") {
        if (value != 64 && value != 32 && value != 16 && value != 8 && value != 4 && value != 2 && value != 1) return fail(ctx, AVK_E_ARG, "%s must be 64, 32, 16, 8, 4, 2 or 1", name);
        (n == "lane_width_one" ? ctx->lane_width_one : (n == "lane_width_two" ? ctx->lane_width_two : ctx->lane_width_three)) = value;
    } else if (n == "lane_head_width") {
        if (value != 0 && value != 64 && value != 32 && value != 16 && value != 8 && value != 4) return fail(ctx, AVK_E_ARG, "lane_head_width must be 0, 64, 32, 16, 8 or 4");
        ctx->lane_head_width = value;
    } else if (n == "hbm_ed_cap") {
        if (value < 0 || value > 1000000) return fail(ctx, AVK_E_ARG, "hbm_ed_cap must be 0..1000000");
        ctx->hbm_ed_cap = value;
    } else if (n == "lane_pairs") {
        ctx->lane_pairs = value ? 1 : 0;
    } else if (n == "lane_min_batch") {
        if (value < 0) return fail(ctx, AVK_E_ARG, "lane_min_batch must not be negative");
        ctx->lane_min_batch = value;
    } else if (n == "lane_node_cap") {
        if (value < 8 || value > 250) return fail(ctx, AVK_E_ARG, "lane_node_cap must be 8..250");
        ctx->lane_node_cap = value;
    } else if (n == "lane_quad") {
        ctx->lane_quad = value ? 1 : 0;
    } else
        return fail(ctx, AVK_E_ARG, "unknown option '%s'", name);
    return 0;
}

int avk_ref_upload(avk_ctx *ctx, uint32_t n_contigs, const uint8_t *const *seqs, const uint64_t *lens) {
    if (!ctx || (n_contigs && (!seqs || !lens))) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->d_ref) {
        AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_ref);
        ctx->d_ref = nullptr;
        if (ctx->d_ref2b) (void)hipFree(ctx->d_ref2b);
        if (ctx->d_refexc) (void)hipFree(ctx->d_refexc);
        ctx->d_ref2b = ctx->d_refexc = nullptr;
    }
    ctx->contig_base.assign(n_contigs, 0);
    ctx->contig_len.assign(n_contigs, 0);
    uint64_t total = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        ctx->contig_base[c] = total;
        ctx->contig_len[c] = lens[c];
        total += lens[c];
    }
    /* any failure leaves the context WITHOUT a reference (later calls then fail with "avk_ref_upload has not been called") */
    hipError_t e = hipMalloc((void **)&ctx->d_ref, total + 64);
    for (uint32_t c = 0; c < n_contigs && e == hipSuccess; ++c)
        if (lens[c]) e = hipMemcpyAsync(ctx->d_ref + ctx->contig_base[c], seqs[c], lens[c], hipMemcpyHostToDevice, ctx->stream);
    const uint64_t n_words = (total + 15) >> 4;
    if (ctx->d_contig_tab) (void)hipFree(ctx->d_contig_tab);
    ctx->d_contig_tab = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_contig_tab, ((size_t)2 * n_contigs + 2) * sizeof(uint64_t));
    if (e == hipSuccess && n_contigs) e = hipMemcpyAsync(ctx->d_contig_tab, ctx->contig_base.data(), (size_t)n_contigs * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_contigs) e = hipMemcpyAsync(ctx->d_contig_tab + n_contigs, ctx->contig_len.data(), (size_t)n_contigs * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_ref2b, (n_words + 80) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_refexc, ((n_words >> 5) + 8) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_ref2b, 0, (n_words + 80) * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_refexc, 0, ((n_words >> 5) + 8) * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) {
        const uint64_t threads = ((n_words + 63) / 64 + 1) * 64;
        hipLaunchKernelGGL(avk_pack_reference, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref, total, ctx->d_ref2b, ctx->d_refexc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->d_ref) (void)hipFree(ctx->d_ref);
        if (ctx->d_ref2b) (void)hipFree(ctx->d_ref2b);
        if (ctx->d_refexc) (void)hipFree(ctx->d_refexc);
        if (ctx->d_contig_tab) (void)hipFree(ctx->d_contig_tab);
        ctx->d_contig_tab = nullptr;
        ctx->d_ref = nullptr;
        ctx->d_ref2b = ctx->d_refexc = nullptr;
        ctx->contig_base.clear();
        ctx->contig_len.clear();
        return fail(ctx, e == hipErrorOutOfMemory ? AVK_E_OOM : AVK_E_HIP, "reference upload failed: %s", hipGetErrorString(e));
    }
    return 0;
}

uint32_t avk_seq_stride(const avk_region_batch *batch, uint64_t r) {
    if (!batch || r >= batch->n_regions) return 0;
    return avk::seq_stride_of(batch, r);
}

/* DESIGN.md "algorithmic bytes per region" (SURVEY.md 8d): in — 2-bit window, 12 B record and 2-bit alleles per call, 16 B header; out — 16 B
 * record, 2 B per call and, when the run writes per-region BASEPAIR groups (with_groups), 16 B for the joint group and 16 B per call type present */
uint64_t avk_algorithmic_bytes_ex(const avk_region_batch *b, int with_groups) {
    if (!b) return 0;
    uint64_t total = 0;
    for (uint64_t r = 0; r < b->n_regions; ++r) {
        uint64_t L = b->end[r] > b->start[r] ? b->end[r] - b->start[r] : 0;
        uint64_t bytes = (L + 3) / 4 + 16 + 16 + (with_groups ? 16 : 0);
        uint32_t types = 0;
        for (int side = 0; side < 2; ++side) {
            uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
            uint32_t cnt = side == 0 ? b->t_cnt[r] : b->q_cnt[r];
            if (off > b->n_variants || cnt > b->n_variants - off) continue;
            for (uint32_t i = 0; i < cnt; ++i) {
                uint64_t v = off + i;
                bytes += 12 + (b->a0_len[v] + 3) / 4 + (b->a1_len[v] + 3) / 4 + 2;
                types |= 1u << (b->var_type[v] & 15);
            }
        }
        if (with_groups) bytes += 16ull * (uint64_t)__builtin_popcount(types);
        total += bytes;
    }
    return total;
}
uint64_t avk_algorithmic_bytes(const avk_region_batch *b) { return avk_algorithmic_bytes_ex(b, 1); }

/* the wide result arrays from the packed form (include/aardvark_amd.h) */
int avk_results_expand(const avk_region_batch *b, const avk_result_batch *packed, avk_result_batch *wide) {
    if (!b || !packed || !wide) return AVK_E_ARG;
    const bool want_region = wide->status || wide->ed_h1 || wide->ed_h2 || wide->n_optima || wide->type_present;
    const bool want_calls = wide->var_expected || wide->var_observed || wide->var_class || wide->var_zyg;
    if ((want_region && !packed->region_packed) || (want_calls && !packed->var_packed)) return AVK_E_ARG;
    if (!want_region && !want_calls) return 0;
    std::atomic<int> bad(0);
    const uint64_t n = b->n_regions, piece = 1u << 16, pieces = (n + piece - 1) / piece;
    std::atomic<uint64_t> next(0);
    AvkPool::get().run(pieces > 1 ? avk_host_threads() : 1, [&](unsigned) {
        for (;;) {
            const uint64_t p = next.fetch_add(1);
            if (p >= pieces) break;
            for (uint64_t r = p * piece; r < n && r < (p + 1) * piece; ++r) {
                uint32_t types = 0;
                for (int side = 0; side < 2; ++side) {
                    const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
                    const uint32_t cnt = side == 0 ? b->t_cnt[r] : b->q_cnt[r];
                    if (off > b->n_variants || cnt > b->n_variants - off) {
                        bad.store(1);
                        continue;
                    }
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const uint64_t v = off + i;
                        if (b->var_type && b->var_type[v] < AVK_N_VARIANT_TYPES) types |= 1u << b->var_type[v];
                        if (!want_calls) continue;
                        const uint8_t x = packed->var_packed[v];
                        if (wide->var_expected) wide->var_expected[v] = avk_vp_expected(x);
                        if (wide->var_observed) wide->var_observed[v] = avk_vp_observed(x);
                        if (wide->var_class) wide->var_class[v] = avk_vp_class(x, side);
                        if (wide->var_zyg) wide->var_zyg[v] = avk_vp_zyg(x);
                    }
                }
                if (!want_region) continue;
                const uint64_t w = packed->region_packed[r];
                if (wide->status) wide->status[r] = avk_rp_status(w);
                if (wide->ed_h1) wide->ed_h1[r] = avk_rp_ed_h1(w);
                if (wide->ed_h2) wide->ed_h2[r] = avk_rp_ed_h2(w);
                if (wide->n_optima) wide->n_optima[r] = avk_rp_n_optima(w);
                if (wide->type_present) wide->type_present[r] = avk_rp_type_present(w, types);
            }
        }
    });
    return bad.load() ? AVK_E_ARG : 0;
}

/* GroupTypeMetrics of one region from its compact BASEPAIR groups and the per-call outputs (include/aardvark_amd.h) */
int avk_group_metrics_from_compact(const avk_region_batch *b, uint64_t r, const avk_result_batch *res, uint32_t *out) {
    const bool bp_packed = res && res->bp_packed && res->bp_spilled;
    if (!b || !res || !out || r >= b->n_regions || !((res->var_expected && res->var_observed) || res->var_packed) || !(res->bp_off || bp_packed) || !res->bp_groups) return AVK_E_ARG;
    const bool wide_calls = res->var_expected && res->var_observed; /* otherwise the packed bytes */
    memset(out, 0, sizeof(uint32_t) * AVK_N_GROUPS * AVK_N_FIELDS);
    uint32_t types = 0;
    uint64_t tot[2][1 + AVK_N_VARIANT_TYPES];
    memset(tot, 0, sizeof(tot));
    for (int side = 0; side < 2; ++side) {
        const uint64_t off = side == 0 ? b->t_off[r] : b->q_off[r];
        const uint32_t cnt = side == 0 ? b->t_cnt[r] : b->q_cnt[r];
        if (off > b->n_variants || cnt > b->n_variants - off) return AVK_E_ARG;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint64_t v = off + i;
            const uint32_t vt = b->var_type[v];
            if (vt >= AVK_N_VARIANT_TYPES) return AVK_E_ARG;
            types |= 1u << vt;
            const uint64_t w = avk::host_edit_distance(b->allele_bytes + b->a0_off[v], b->a0_len[v], b->allele_bytes + b->a1_off[v], b->a1_len[v]); /* Variant::alt_ed */
            /* the query entries are stored toggled (compare_benchmark.rs:109-123): scored as truth they expected var_observed and observed var_expected */
            const uint32_t ea = wide_calls ? res->var_expected[v] : avk_vp_expected(res->var_packed[v]), oa = wide_calls ? res->var_observed[v] : avk_vp_observed(res->var_packed[v]);
            const uint32_t exp = side == 0 ? ea : oa, obs = side == 0 ? oa : ea;
            const int f_gt_tp = side ? AVK_F_GT_QUERY_TP : AVK_F_GT_TRUTH_TP, f_gt_fn = side ? AVK_F_GT_QUERY_FP : AVK_F_GT_TRUTH_FN, f_gt_fn_gt = side ? AVK_F_GT_QUERY_FP_GT : AVK_F_GT_TRUTH_FN_GT;
            const int f_hap_tp = side ? AVK_F_HAP_QUERY_TP : AVK_F_HAP_TRUTH_TP, f_hap_fn = side ? AVK_F_HAP_QUERY_FP : AVK_F_HAP_TRUTH_FN;
            const int f_w_tp = side ? AVK_F_WHAP_QUERY_TP : AVK_F_WHAP_TRUTH_TP, f_w_fn = side ? AVK_F_WHAP_QUERY_FP : AVK_F_WHAP_TRUTH_FN;
            for (uint32_t g : {0u, 1u + vt}) { /* GroupMetrics::add_truth_zygosity (grouped_metrics.rs:183-227) on the joint group and the type's */
                uint32_t *G = out + g * AVK_N_FIELDS;
                G[f_hap_tp] += obs;
                G[f_hap_fn] += exp - obs;
                G[f_w_tp] += (uint32_t)(obs * w);
                G[f_w_fn] += (uint32_t)((exp - obs) * w);
                if (exp == obs) G[f_gt_tp] += 1;
                else {
                    G[f_gt_fn] += 1;
                    if (obs > 0) G[f_gt_fn_gt] += 1;
                }
            }
            const uint32_t z = b->var_zyg[v];
            const uint64_t cz = z == AVK_ZYG_HOM_ALT ? 2 : ((z >= AVK_ZYG_UNPHASED_HET && z <= AVK_ZYG_PHASED_HET10) ? 1 : 0);
            const uint64_t raw = b->var_raw_space ? b->var_raw_space[v] : (b->a0_len[v] > b->a1_len[v] ? b->a0_len[v] : b->a1_len[v]);
            tot[side][0] += cz * raw;
            tot[side][1 + vt] += cz * raw;
        }
    }
    /* BASEPAIR from the compact groups (joint, then the region's call types in type order); RECORD_BP from them and the totals (waffle_solver.rs:455-522) */
    uint32_t simple[4] = {0, 0, 0, 0}; /* the packed form's word: every group of the region is this one */
    uint32_t lo = 0, hi = 0;
    bool one_for_all = false;
    if (bp_packed) {
        const uint32_t w = res->bp_packed[r], n_groups = 1u + (uint32_t)__builtin_popcount(types);
        if (avk_bp_is_spilled(w)) {
            lo = avk_bp_spill_index(w), hi = lo + n_groups;
            if (hi > res->bp_spilled[0]) return AVK_E_ARG;
        } else {
            for (int i = 0; i < 4; ++i) simple[i] = avk_bp_counter(w, i);
            one_for_all = true, hi = n_groups;
        }
    } else
        lo = res->bp_off[r], hi = res->bp_off[r + 1];
    uint32_t k = lo;
    for (uint32_t left = 1u | (types << 1); left; left &= left - 1, ++k) {
        if (k >= hi) return AVK_E_ARG; /* the region owns no groups (it failed validation) or fewer than its call types */
        const uint32_t g = (uint32_t)__builtin_ctz(left);
        uint32_t *G = out + g * AVK_N_FIELDS;
        const uint32_t *bp = one_for_all ? simple : res->bp_groups + 4 * (size_t)k;
        for (int i = 0; i < 4; ++i) G[AVK_F_BP_TRUTH_TP + i] = bp[i];
        G[AVK_F_RBP_TRUTH_TP] = (uint32_t)(2 * tot[0][g] - bp[1]);
        G[AVK_F_RBP_TRUTH_FN] = bp[1];
        G[AVK_F_RBP_QUERY_TP] = (uint32_t)(2 * tot[1][g] - bp[3]);
        G[AVK_F_RBP_QUERY_FP] = bp[3];
    }
    return k == hi ? 0 : AVK_E_ARG;
}

static int upload_internal(avk_ctx *ctx, const avk_region_batch *batch, bool pairs_mode, avk_dev_batch **out) {
    if (!ctx || !batch || !out) return AVK_E_ARG;
    *out = nullptr;
    if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->device_pack) return upload_device_packed(ctx, batch, nullptr, pairs_mode, out);
    const bool timing = getenv("AVK_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_start = now();
    avk_dev_batch *db = new avk_dev_batch();
    db->n_regions = batch->n_regions;
    db->n_variants_host = batch->n_variants;
    /* sequence slots are laid out by the library: prefix of 5 * stride */
    std::vector<uint64_t> seq_off(batch->n_regions);
    std::vector<uint32_t> seq_stride(batch->n_regions);
    uint64_t seq_total = 0;
    {
        const uint64_t nr = batch->n_regions;
        size_t nt = avk_usable_cpus();
        if (nt > 16) nt = 16;
        if (nt > nr / 65536 + 1) nt = (size_t)(nr / 65536 + 1);
        auto part = [&](size_t t) {
            for (uint64_t r = nr * t / nt; r < nr * (t + 1) / nt; ++r) seq_stride[r] = avk::seq_stride_of(batch, r);
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(part, t);
        part(0);
        for (auto &th : pool) th.join();
    }
    for (uint64_t r = 0; r < batch->n_regions; ++r) {
        seq_off[r] = seq_total;
        seq_total += 5ull * seq_stride[r];
    }
    db->seq_total = seq_total;
    std::string err;
    int rc = avk::pack_batch(batch, ctx->contig_base, ctx->contig_len, seq_off.data(), seq_stride.data(), &db->host, &err, 0, (uint32_t)ctx->lane_max_est, ctx->lane_pairs != 0 && !pairs_mode, (uint32_t)ctx->het_search_min);
    if (rc) {
        delete db;
        return fail(ctx, rc, "%s", err.c_str());
    }
    const uint64_t n = db->n_regions, nv = db->host.variants.size();
    db->n_variants_dev = nv;
    const auto t_packed = now();
    if (pairs_mode) { /* solve_merge_region's pre-checks (merge_solver.rs:119-147, :211-223) */
        for (uint64_t r = 0; r < n; ++r) {
            AvkDevRegion &dr = db->host.regions[r];
            if ((dr.pre_status & 0xFFFFu) == AVK_ST_INVALID_INPUT) continue;
            if (db->host.zyg_flags[r] & 1) dr.pre_status = AVK_ST_BAD_ZYGOSITY;                           /* Unknown: delta computation bails */
            else if (db->host.delta_t[r] != db->host.delta_q[r]) dr.pre_status = AVK_PRE_SKIP_OK;        /* different net length: not exact, optimizer not run */
            else if (db->host.zyg_flags[r] & 2) dr.pre_status = AVK_ST_BAD_ZYGOSITY;                      /* optimizer would hit assert_eq! */
            else dr.pre_status = 0;
        }
    }
#define AVK_TRY(x)                \
    do {                          \
        int rc_ = (x);            \
        if (rc_) {                \
            free_batch_buffers(db); \
            delete db;            \
            return rc_;           \
        }                         \
    } while (0)
    AVK_TRY(dev_alloc(ctx, &db->d_regions, n));
    AVK_TRY(dev_alloc(ctx, &db->d_blob, db->host.blob.size()));
    AVK_TRY(dev_alloc(ctx, &db->d_region_out, n * 4));
    AVK_TRY(dev_alloc(ctx, &db->d_var_out, nv));
    AVK_TRY(dev_alloc(ctx, &db->d_seqlen, n * 5));
    AVK_TRY(dev_alloc(ctx, &db->d_tally, (size_t)AVK_TALLY_STRIDE));
    AVK_TRY(dev_alloc(ctx, &db->d_partials, (size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES));
    AVK_TRY(dev_alloc(ctx, &db->d_counters, (size_t)AVK_N_COUNTERS));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow, n + 1024));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow2, n + 1));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow3, 3 * (n + 1) + 1024)); /* (the lanes' hand-backs: a segment per chain, per head launch and for the looked-up pairs) */
    AVK_TRY(dev_alloc(ctx, &db->d_overflow4, n + 1));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow5, n + 1));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow6, n + 1));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow7, n + 1));
    AVK_TRY(dev_alloc(ctx, &db->d_overflow8, n + 1));
    AVK_TRY(dev_alloc(ctx, &db->d_notwide, n + 2));
#undef AVK_TRY
    { /* compact BASEPAIR groups: 1 + the region's call types each (none for regions that fail validation), as the device packer counts them */
        std::vector<uint32_t> bp_off(n + 1, 0);
        for (uint64_t r = 0; r < n; ++r) {
            const uint32_t ps = db->host.regions[r].pre_status;
            bp_off[r + 1] = bp_off[r] + ((ps & 0xFFFFu) ? 0u : 1u + (uint32_t)__builtin_popcount(ps >> 16));
        }
        db->n_bp_groups = bp_off[n];
        hipError_t eb = hipMalloc((void **)&db->d_bp_off, (n + 2) * sizeof(uint32_t));
        if (eb == hipSuccess) eb = hipMemcpy(db->d_bp_off, bp_off.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (eb != hipSuccess) {
            free_batch_buffers(db);
            delete db;
            return fail(ctx, AVK_E_HIP, "batch upload failed: %s", hipGetErrorString(eb));
        }
    }
    /* work order: the regions predicted to outgrow the small LDS slice first (solo waves take them), then the
     * rest; within each part the regions with the most variants (the expensive searches) are dealt first, so
     * they overlap with the bulk instead of forming the tail */
    const auto t_alloc = now();
    std::vector<uint32_t> order;
    db->plan = avk::plan_work_order(db->host, avk::bulk_slice_bytes((uint64_t)ctx->lds_bytes_per_wave), (uint32_t)ctx->lds_ed_cap, (uint64_t)ctx->lds2_bytes_per_wave,
                                    (uint32_t)ctx->lds2_ed_cap, pairs_mode && !ctx->pair_classes ? 0u : (uint32_t)ctx->solo_min_variants, 50, &order, (uint32_t)ctx->class_c_nodes_x2,
                                    ctx->lane_kernel && ctx->use_packed_reference && ctx->d_ref2b ? (uint64_t)ctx->lane_min_regions : 0xFFFFFFFFull, (uint32_t)ctx->lane_max_calls,
                                    (uint64_t)ctx->lane_min_batch, ctx->lane_stripe ? (uint32_t)ctx->lane_head_width : 0u, (uint32_t)ctx->lane_head_est, (uint32_t)ctx->het_search_min,
                                    (uint64_t)ctx->class_c_below);
    const auto t_plan = now();
    hipError_t e = hipSuccess;
    /* the records go up in work order: a wave reads record k of its launch's range, no index list in between */
    const avk::PodVec<AvkDevRegion> sorted = avk::regions_in_work_order(db->host, order);
    avk::PodVec<uint32_t> fast;
    if (db->plan.n_fast_total) {
        fast = avk::build_fast_records(db->host, order, db->plan, db->fast_word_base, db->fast_tiles);
        e = hipMalloc((void **)&db->d_fast, fast.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(db->d_fast, fast.data(), fast.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    }
    if (n && e == hipSuccess) e = hipMemcpyAsync(db->d_regions, sorted.data(), n * sizeof(AvkDevRegion), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db->d_blob, db->host.blob.data(), db->host.blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        free_batch_buffers(db);
        delete db;
        return fail(ctx, AVK_E_HIP, "batch upload failed: %s", hipGetErrorString(e));
    }
    if (timing)
        fprintf(stderr, "avk upload: pack %.3f ms, alloc %.3f ms, plan %.3f ms, copy %.3f ms\n", ms(t_start, t_packed), ms(t_packed, t_alloc), ms(t_alloc, t_plan),
                ms(t_plan, now()));
    *out = db;
    return 0;
}

int avk_batch_upload(avk_ctx *ctx, const avk_region_batch *batch, avk_dev_batch **out) { return upload_internal(ctx, batch, false, out); }

void avk_batch_free(avk_ctx *ctx, avk_dev_batch *db) {
    if (!db) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (ctx && db->dev_packed) { /* back to the context's pool: nothing to wait for */
        release_pooled(ctx, db);
        delete db;
        return;
    }
    /* a large batch holds a dozen device buffers and about a gigabyte of host records: releasing them takes 0.1 s, which a tool that
     * works through its batches one after the other should not wait for.  One release runs behind the caller at a time. */
    if (ctx && db->n_regions >= 65536) {
        std::lock_guard<std::mutex> lock(ctx->reaper_mutex);
        if (ctx->reaper.joinable()) ctx->reaper.join();
        const int device = ctx->device;
        ctx->reaper = std::thread([db, device] {
            (void)hipSetDevice(device);
            free_batch_buffers(db);
            delete db;
        });
        return;
    }
    free_batch_buffers(db);
    delete db;
}

/* tile width of a lane launch (options lane_width_one / lane_width_two: 64, 32 or 16 records per wave at a time) */
static uint32_t lane_width_log2(const avk_ctx *ctx, uint32_t maxv) {
    const int64_t w = maxv > 2 ? ctx->lane_width_three : (maxv > 1 ? ctx->lane_width_two : ctx->lane_width_one);
    return w <= 1 ? 0u : (w <= 2 ? 1u : (w <= 4 ? 2u : (w <= 8 ? 3u : (w <= 16 ? 4u : (w <= 32 ? 5u : 6u)))));
}
/* LDS bytes of a one-wave workgroup of the lane kernel (0: does not fit) and the grid that fills the machine: the per-lane arrays of
 * `width` lanes plus a tally of its own; as many workgroups per CU as the LDS and the wave slots hold */
static uint32_t head_width_log2(const avk_ctx *ctx, uint32_t head_regions) {
    int64_t w = ctx->lane_head_width;
    if (ctx->lane_head_auto && w > 4) { /* lanes that diverge take turns: a head that cannot fill the machine's lane waves 16 wide is spread over more, narrower waves */
        const int64_t fit = head_regions < 8192u ? 4 : (head_regions < 24576u ? 8 : 16);
        if (fit < w) w = fit;
    }
    return w <= 1 ? 0u : (w <= 2 ? 1u : (w <= 4 ? 2u : (w <= 8 ? 3u : (w <= 16 ? 4u : (w <= 32 ? 5u : 6u)))));
}
/* which of the two kernels a lane launch runs: four lanes per region when the launch is narrow (option lane_quad) */
static bool lane_launch_is_quad(const avk_ctx *ctx, const avk::lane::LaneArgs &la) { return ctx->lane_quad && la.lanes_log2 <= 4; }
static size_t lane_launch_geometry(const avk_ctx *ctx, const avk::lane::LaneArgs &la, uint32_t *grid) {
    const uint32_t rows = (1 + 2 * (la.nm - 1)) * (la.W + 1) + ((lane_launch_is_quad(ctx, la) ? 4 : 3) + 2 * la.pool) * ((2 * la.ed_max + 2 + 3) / 4) + la.qcap + (la.nm == 2 ? 4 : 8); /* lane_rows / quad_rows */
    const size_t lds = (size_t)rows * (4u << la.lanes_log2) + 288 * 4;
    uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
    if (per_cu < 1) return 0;
    uint32_t cap = la.lanes_log2 < 4 ? 32u : (uint32_t)(ctx->lane_waves_per_cu > 0 ? ctx->lane_waves_per_cu : 12); /* very narrow tiles: every wave slot */
    if (la.nm > 4 && ctx->lane_waves_three > 0) cap = (uint32_t)ctx->lane_waves_three; /* the three-call class: 17 KB per wave */
    if (per_cu > cap) per_cu = cap;
    const uint32_t claims = la.n_tiles * (64u >> la.lanes_log2);
    uint32_t g = (uint32_t)ctx->n_cus * per_cu;
    if (g > claims) g = claims;
    *grid = g;
    return lds;
}
/* a launch of a lane class: one lane per region, or four (lane_launch_is_quad) */
/* done: an event that fires when THIS launch ends (bound to the launch's own completion signal: an event recorded behind the launch is a packet of its own in the
 * stream, and the next launch of the chain started 55-60 us later for it) */
static void launch_lane_class(const avk_ctx *ctx, uint32_t grid, size_t lds, hipStream_t s, const AvkKernelArgs &f, const avk::lane::LaneArgs &la, hipEvent_t done = nullptr) {
    if (lane_launch_is_quad(ctx, la)) {
        if (lds * 8 > 160 * 1024) hipExtLaunchKernelGGL(avk_quad_kernel_wide_regs, dim3(grid), dim3(64), lds, s, nullptr, done, 0, f, la); /* fewer than eight workgroups per CU by LDS: registers to spare */
        else hipExtLaunchKernelGGL(avk_quad_kernel, dim3(grid), dim3(64), lds, s, nullptr, done, 0, f, la);
    }
    else hipExtLaunchKernelGGL(avk_lane_kernel, dim3(grid), dim3(64), lds, s, nullptr, done, 0, f, la);
}

/* The table of avk_pairs.inl for this max_branch_factor, on `s`: sixteen probe regions through the lane kernel, outputs redirected into the table.
 * Words of d_pair_aux: [0, 768) probe records, [768, 784) probe reference, [784, 792) its exception bitmap (zeros), [792, 812) BASEPAIR offsets 0, 2, .. 32,
 * [812] tile counter, [813] overflow count, [816, 880) overflow list, [896, 896 + 2 * AVK_TALLY_STRIDE) a partial tally nobody reads. */
static int ensure_pair_table(avk_ctx *ctx, uint32_t max_branch_factor, hipStream_t s) {
    enum { AUX_WORDS = 896 + 2 * AVK_TALLY_STRIDE };
    if (!ctx->d_pair_tab) {
        AVK_HIP(ctx, hipMalloc((void **)&ctx->d_pair_tab, sizeof(avk::pairs::PairTable)));
        AVK_HIP(ctx, hipMalloc((void **)&ctx->d_pair_aux, (size_t)AUX_WORDS * 4));
        std::vector<uint32_t> aux(AUX_WORDS, 0);
        avk::pairs::pair_probe_records(aux.data(), aux.data() + 768);
        for (uint32_t k = 0; k <= avk::pairs::N_SIG; ++k) aux[792 + k] = 2 * k;
        AVK_HIP(ctx, hipMemcpy(ctx->d_pair_aux, aux.data(), (size_t)AUX_WORDS * 4, hipMemcpyHostToDevice));
        ctx->pair_tab_mbf = -1;
    }
    if (ctx->pair_tab_mbf == (int64_t)max_branch_factor) return AVK_E_OK;
    AVK_HIP(ctx, hipMemsetAsync(ctx->d_pair_tab, 0xFF, sizeof(avk::pairs::PairTable), s));
    AVK_HIP(ctx, hipMemsetAsync(ctx->d_pair_aux + 812, 0, (size_t)(AUX_WORDS - 812) * 4, s));
    AvkKernelArgs f;
    memset(&f, 0, sizeof(f));
    f.ref_2bit = ctx->d_pair_aux + 768;
    f.ref_exc = ctx->d_pair_aux + 784;
    f.n_regions = avk::pairs::N_SIG;
    f.max_branch_factor = max_branch_factor;
    f.region_out = &ctx->d_pair_tab->region[0][0];
    f.var_out = &ctx->d_pair_tab->var[0][0];
    f.group_metrics = &ctx->d_pair_tab->gm[0][0];
    f.bp_off = ctx->d_pair_aux + 792;
    f.bp_out = &ctx->d_pair_tab->bp[0][0];
    f.tally = (uint64_t *)(ctx->d_pair_aux + 896);
    f.overflow_list = ctx->d_pair_aux + 816;
    f.overflow_count = ctx->d_pair_aux + 813;
    avk::lane::LaneArgs la;
    memset(&la, 0, sizeof(la));
    const AvkFastClass &cl = AVK_FAST_CLASS[0];
    la.recs = ctx->d_pair_aux;
    la.rec_words = AVK_FAST_WORDS_OF(1);
    la.n_tiles = 1;
    la.tile_counter = ctx->d_pair_aux + 812;
    la.W = cl.W, la.nm = 2, la.ed_max = cl.ed_max, la.qcap = cl.qcap;
    la.lanes_log2 = 6;
    la.max_nodes = 250;
    uint32_t grid = 0;
    const size_t lds = lane_launch_geometry(ctx, la, &grid);
    if (!ctx->lane_attr_set) {
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_lane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_quad_kernel_wide_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->lane_attr_set = true;
    }
    hipLaunchKernelGGL(avk_lane_kernel, dim3(1), dim3(64), lds, s, f, la);
    AVK_HIP(ctx, hipGetLastError());
    ctx->pair_tab_mbf = (int64_t)max_branch_factor;
    return AVK_E_OK;
}

/* workgroups (x 4 waves x ws_bytes_per_wave of HBM) of the main HBM-tier launch of a batch that has lane launches.  Every wave of the grid needs a slice, and fresh device
 * memory is scrubbed when it is handed out (75 ms per GB): n_cus x 3 workgroups were 3 of the 4.8 GB a process's first whole-genome call waited 0.39 s for
 * (profiles/r05_first_solve.txt); the step does not notice the difference (profiles/r05_quad_sweeps.txt) */
#define AVK_HBM_BLOCKS_BESIDE_LANES 192u
static int run_internal(avk_ctx *ctx, avk_dev_batch *db, const avk_compare_config *cfg, void *tally_dev, uint32_t mode) {
    if (!ctx || !db || !cfg) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = db->n_regions;
    if (ctx->lds2_bytes_per_wave * 4 > 160 * 1024) return fail(ctx, AVK_E_ARG, "lds2_bytes_per_wave too large");
    /* sequences: allocate the output arena on first use */
    if (cfg->enable_sequences && !db->d_seq) {
        int rc = db->dev_packed ? pool_alloc_t(ctx, db, &db->d_seq, (size_t)db->seq_total + 16) : dev_alloc(ctx, &db->d_seq, (size_t)db->seq_total + 16);
        if (rc) return rc;
    }
    /* up to four launches, one per workspace tier; each consumes the overflow list of the one
     * before it (its length is read on the device, so nothing comes back to the host in between) */
    const bool use[4] = {ctx->lds_bytes_per_wave > 0, ctx->lds2_bytes_per_wave > 0, ctx->ws_bytes_per_wave > 0, ctx->big_ws_bytes > 0};
    /* which tiers get a launch of their own: the large LDS slices normally only serve the solo launch (their
     * overflow pass runs one workgroup per CU and measured slower than handing the overflow to the HBM tier,
     * which runs twice the waves), and the big HBM slices are claimed in place by the tier-2 launch */
    const bool launch[4] = {use[0], use[1] && (ctx->lds2_overflow_pass || !use[0]), use[2], use[3] && !use[2]};
    int last = -1;
    for (int t = 0; t < 4; ++t)
        if (launch[t]) last = t;
    if (last < 0) return fail(ctx, AVK_E_ARG, "every workspace tier is disabled");
    /* The lane-per-region kernel takes the fast segments at the end of the work order (WorkPlan::fast_base); what it cannot finish it
     * appends to the list the first HBM launch reads.  Without such a launch, with sequence output (the haplotype bytes are never
     * materialised there) or with the exact-match shortcut the wave-per-region bulk launch covers those records itself. */

    /* which classes go to the lanes was decided with the work order (plan_work_order, option lane_min_regions at upload time) */
    const int lane_k = AVK_FAST_CLASSES;
    const uint32_t n_lane_regions = db->plan.n_fast_total;
    /* the lanes read the 2-bit reference only: with use_packed_reference off (a.ref_2bit == NULL) the wave-per-region kernels take every record */
    const bool use_fast = ctx->lane_kernel && launch[0] && !launch[1] && launch[2] && !launch[3] && db->d_fast && n_lane_regions && !cfg->enable_sequences &&
                          !cfg->enable_exact_shortcut && n && ctx->use_packed_reference && ctx->d_ref2b;
    const uint32_t n_fast = use_fast ? n_lane_regions : 0u;
    /* the wave-cooperative kernel of avk_wide.inl takes class C and what the three-call lane class hands back ahead of the HBM-tier launches (which get what it
     * cannot take); like the lanes it reads the 2-bit reference and writes no sequences */
    const bool use_wide = ctx->wide_kernel && launch[0] && !launch[1] && launch[2] && !launch[3] && !cfg->enable_sequences && !cfg->enable_exact_shortcut && n &&
                          ctx->use_packed_reference && ctx->d_ref2b;
    avk::wide::WideArgs wa;
    wa.lds_words = (uint32_t)(ctx->wide_lds_bytes / 4);
    wa.skip_static = 0;
    if (use_wide && !ctx->wide_attr_set) {
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_wide_kernel_lazy, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        ctx->wide_attr_set = true;
    }
    if (db->dev_packed && !use_fast) { /* the wave-per-region launches take every record: write the ones the packer left for later */
        const int rf = ensure_all_records(ctx, db, ctx->stream);
        if (rf) return rf;
    }
    /* geometry */
    const uint32_t waves_per_block = 4;
    uint64_t want_waves = (uint64_t)ctx->n_cus * (uint64_t)ctx->waves_per_cu;
    if (want_waves > n - n_fast) want_waves = n - n_fast;
    uint32_t blocks = (uint32_t)((want_waves + waves_per_block - 1) / waves_per_block);
    if (blocks == 0) blocks = 1;
    /* the HBM launches: no more workgroups than can be resident (three per CU at the kernel's register count) — later ones would find
     * the work list empty anyway, and every wave of the grid needs a slice of HBM (fresh device memory is scrubbed when it is handed
     * out: 13 GB of workspaces cost 0.25 s at the first call of a process, 4 GB a third of that) */
    uint32_t hbm_blocks = (uint32_t)ctx->n_cus * 3u;
    if (use_fast && n - n_fast <= 16384 && hbm_blocks > AVK_HBM_BLOCKS_BESIDE_LANES) hbm_blocks = AVK_HBM_BLOCKS_BESIDE_LANES; /* lane launches that leave the other kernels a few thousand regions leave this tier a few dozen (a genome: 27) */
    if (hbm_blocks > blocks) hbm_blocks = blocks;
    /* the HBM solo launch runs beside the main stream's HBM launch: its (at most 64) workgroups have slices of their own, after the others */
    uint32_t hbm_solo_max = (uint32_t)(ctx->hbm_solo_blocks > 0 ? ctx->hbm_solo_blocks : 128);
    uint32_t hbm_early_max = (uint32_t)(ctx->hbm_early_blocks > 0 ? ctx->hbm_early_blocks : 64); /* workgroups of the launch behind the three-call lane class (its hand-backs), slices of their own too */
    /* the per-wave slice: the option, or what the packer's prediction asks for (device-packed batches of large windows: upload_device_packed); large
     * slices mean fewer waves per launch (option ws_budget_bytes) and more of the shared big slices for what still overflows */
    const int64_t ws_bytes = db->ws_bytes_eff > ctx->ws_bytes_per_wave ? db->ws_bytes_eff : ctx->ws_bytes_per_wave;
    const int64_t big_waves = ws_bytes > ctx->ws_bytes_per_wave && ctx->big_waves < 64 ? 64 : ctx->big_waves;
    {
        const double total = (double)(hbm_blocks + hbm_solo_max + hbm_early_max) * waves_per_block * (double)ws_bytes;
        if (total > (double)ctx->ws_budget_bytes) {
            const double f = (double)ctx->ws_budget_bytes / total;
            auto shrink = [&](uint32_t b) { const uint32_t x = (uint32_t)(b * f); return x < 8 ? (b < 8 ? b : 8u) : x; };
            hbm_blocks = shrink(hbm_blocks), hbm_solo_max = shrink(hbm_solo_max), hbm_early_max = shrink(hbm_early_max);
        }
    }
    const uint32_t n_waves = hbm_blocks * waves_per_block;
    const size_t ws_need = (size_t)(n_waves + (hbm_solo_max + hbm_early_max) * waves_per_block) * (size_t)ws_bytes;
    const auto t_ws = std::chrono::steady_clock::now();
    const bool ws_grows = ws_need > ctx->ws_alloc;
    if (ws_need > ctx->ws_alloc) {
        if (ctx->d_ws) {
            AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_ws);
            ctx->d_ws = nullptr;
            ctx->ws_alloc = 0;
        }
        AVK_HIP(ctx, hipMalloc((void **)&ctx->d_ws, ws_need + 256));
        ctx->ws_alloc = ws_need;
    }
    const uint32_t big_blocks = (uint32_t)((big_waves + waves_per_block - 1) / waves_per_block);
    const int64_t big_bytes = db->big_bytes_eff > ctx->big_ws_bytes && ctx->big_ws_bytes > 0 ? db->big_bytes_eff : ctx->big_ws_bytes;
    const size_t big_need = (size_t)big_blocks * waves_per_block * (size_t)big_bytes;
    if (big_need > ctx->big_alloc) {
        if (ctx->d_big) {
            AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_big);
            ctx->d_big = nullptr;
            ctx->big_alloc = 0;
        }
        AVK_HIP(ctx, hipMalloc((void **)&ctx->d_big, big_need + 256));
        ctx->big_alloc = big_need;
    }

    if (getenv("AVK_TIMING") && (ws_grows || big_need > 0))
        fprintf(stderr, "avk run: workspaces (%.1f GB per-wave slices%s, %.1f GB shared slices) %.3f ms\n", (double)ws_need / 1e9, ws_grows ? ", allocated now" : "",
                (double)big_need / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ws).count());
    if (!db->scratch_clean) { /* normally left clean by avk_tally_reduce of the previous call */
        AVK_HIP(ctx, hipMemsetAsync(db->d_partials, 0, (size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES * sizeof(uint64_t), ctx->stream));
        AVK_HIP(ctx, hipMemsetAsync(db->d_counters, 0, AVK_N_COUNTERS * sizeof(uint32_t), ctx->stream));
    }
    db->scratch_clean = false;
    if (ctx->emit_bp_groups && mode == 0 && !db->d_bp && db->d_bp_off) {
        int rc = db->dev_packed ? pool_alloc_t(ctx, db, &db->d_bp, (size_t)db->n_bp_groups * 4 + 4) : dev_alloc(ctx, &db->d_bp, (size_t)db->n_bp_groups * 4 + 4);
        if (rc) return rc;
    }
    if (ctx->emit_group_metrics && !db->d_gm) {
        int rc = db->dev_packed ? pool_alloc_t(ctx, db, &db->d_gm, (size_t)n * AVK_N_GROUPS * AVK_N_FIELDS + 4) : dev_alloc(ctx, &db->d_gm, (size_t)n * AVK_N_GROUPS * AVK_N_FIELDS);
        if (rc) return rc;
    }

    AvkKernelArgs a;
    memset(&a, 0, sizeof(a));
    a.regions = db->d_regions;
    a.blob = db->d_blob;
    a.ref_bytes = ctx->d_ref;
    a.ref_2bit = ctx->use_packed_reference ? ctx->d_ref2b : nullptr;
    a.ref_exc = ctx->d_refexc;
    a.n_regions = (uint32_t)n;
    a.max_branch_factor = cfg->max_branch_factor;
    a.enable_exact_shortcut = cfg->enable_exact_shortcut ? 1u : 0u;
    a.mode = mode;
    a.tier[0].ws_bytes = (uint64_t)ctx->lds_bytes_per_wave;
    a.tier[0].ed_cap = (uint32_t)ctx->lds_ed_cap;
    a.tier[1].ws_bytes = (uint64_t)ctx->lds2_bytes_per_wave;
    a.tier[1].ed_cap = (uint32_t)ctx->lds2_ed_cap;
    a.tier[2].ws_bytes = (uint64_t)ws_bytes;
    a.tier[2].ed_cap = ctx->hbm_ed_cap > 0 ? ((uint32_t)ctx->hbm_ed_cap | AVK_CAP_BOUND_ONLY) : 0u;
    a.tier[3].ws_bytes = (uint64_t)big_bytes;
    a.tier[3].ed_cap = 0;
    a.region_out = db->d_region_out;
    a.group_metrics = ctx->emit_group_metrics ? db->d_gm : nullptr;
    a.var_out = db->d_var_out;
    a.seq_bytes = cfg->enable_sequences ? db->d_seq : nullptr;
    a.seq_len = cfg->enable_sequences ? db->d_seqlen : nullptr;
    a.tally = db->d_partials;
    if (ctx->emit_bp_groups && mode == 0 && db->d_bp) {
        a.bp_off = db->d_bp_off;
        a.bp_out = db->d_bp;
    }
    if (db->dev_packed && !db->records_full) { /* (only the launches for handed-back regions ever meet an index >= lazy_from) */
        a.lazy_dp = db->d_dp_args;
        a.lazy_from = db->lazy_from;
    }

    if (cfg->max_branch_factor == 0) return fail(ctx, AVK_E_ARG, "max_branch_factor must be greater than 0 (query_optimizer.rs:177)");

    if (!ctx->lds_attr_set) {
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_region_kernel_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_region_kernel_lds_lazy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->lds_attr_set = true;
    }
    const bool timed = ctx->timing_events != 0;
    if (timed) AVK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    ctx->ev_lane_valid = false;
    const uint32_t big_slots = use[2] && use[3] ? (uint32_t)(big_waves < 128 ? big_waves : 128) : 0u;
    const uint32_t *list = nullptr, *count = nullptr; /* first launch: the records themselves are in work order */
    uint32_t *lists[4] = {db->d_overflow, db->d_overflow2, db->d_overflow3, db->d_overflow4};
    int nlist = 0;
    bool solo_pending = false, hbm_solo_pending = false, deferred_pending = false, early_pending = false, hbm_shared = false, wide_x_pending = false, chains = false;
    hipEvent_t chain_join[4] = {nullptr, nullptr, nullptr, nullptr}; /* handback_chains: the ends of the lane streams' chains, and the launch that follows them at the end of the step */
    int n_chain_join = 0;
    AvkKernelArgs chain_d;
    uint32_t chain_dblocks = 0;
    memset(&chain_d, 0, sizeof(chain_d));
    uint32_t hbm_shared_base = 0; /* records of class C ahead of the shared list (a team launch has them) */
    for (int t = 0; t < 4 && n; ++t) {
        if (!launch[t]) continue;
        a.pass_tier = (uint32_t)t;
        a.work_list = list;
        a.work_base = 0;
        a.n_work_dev = count;
        a.work_counter = db->d_counters + 256 * t;
        if (t != last) {
            a.overflow_list = lists[nlist];
            a.overflow_count = db->d_counters + 1024 + 16 * nlist;
        } else {
            a.overflow_list = nullptr;
            a.overflow_count = nullptr;
        }
        const bool first_launch = list == nullptr; /* ev0 sits right before it */
        if (t >= 2) { /* the HBM launches read the list the LDS solo launch appends to; the tier-1 launch does not */
            if (deferred_pending && chains) { /* what the chains' launches of avk_wide.inl left: ahead of the waits for the long solo launches, which it does not depend on */
        for (int k = 0; k < n_chain_join; ++k) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, chain_join[k], 0));
        hipLaunchKernelGGL(avk_region_kernel_lds_lazy, dim3(chain_dblocks), dim3(256), (size_t)waves_per_block * (size_t)ctx->lds_bytes_per_wave, ctx->stream, chain_d);
        AVK_HIP(ctx, hipGetLastError());
    }
    if (solo_pending) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            solo_pending = false;
            /* the tier-3 launch of its own (big_slots == 0) reads the list the HBM solo launch appends to */
            if (t == 3 && hbm_solo_pending) {
                AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join2, 0));
                hbm_solo_pending = false;
            }
        }
        a.n_work = (uint32_t)n - (t == 0 ? n_fast : 0u);
        a.high_priority = 0;
        a.esc_bytes = 0;
        a.esc_enabled = 0;
        /* pairs of a merge batch: the bulk launch is long (200,000 regions beside 10 M in the lanes) and its 40 KB workgroups do not all become resident while the
         * lane launches keep taking the LDS that comes free; a statically dealt share would wait for workgroups that start when the lanes are done
         * (12 or 22 ms per whole-genome merge, from call to call) — everything is claimed there */
        /* the same for every short list — the bulk beside the lane classes (a few thousand regions) and the overflow lists of the later tiers: same step,
         * fewer slow calls (tools/gpu_static_ab.py); a static share is for the first launch of a batch that goes through the wave-per-region kernels as a whole */
        a.static_pct = mode == 0 && t == 0 && a.n_work >= 16384u ? (uint32_t)ctx->static_pct : 0u; /* (50,000 single-call regions through the bulk: 0.40 ms with the static share, 0.43 without) */
        a.n_shards = 8;
        a.claim = (uint32_t)ctx->claim;
        if (t == 0) {
            a.hbm_ws = nullptr;
            /* Solo launches: the regions the host predicted to outgrow the small slice are solved AT THE SAME TIME, on the
             * side stream, launched just before the bulk, so the long searches overlap with the bulk instead of forming
             * the tail of a later launch:
             *   class B (WorkPlan::n_hard): one-wave workgroups with a tier-1 LDS slice each; a solo workgroup's LDS
             *           displaces exactly one bulk workgroup;
             *   class C (WorkPlan::n_hbm):  predicted to outgrow tier 1 too: a small HBM-tier launch (its registers
             *           displace about two bulk workgroups per workgroup).
             * Both hand what they cannot hold to the list the first HBM launch of the main stream reads. */
            const bool solo_ok = ctx->solo_blocks_max && blocks >= 8;
            const bool hbm_solo_ok = solo_ok && launch[2] && db->plan.n_hbm;
            const uint32_t n_c = hbm_solo_ok ? db->plan.n_hbm : 0u;          /* regions of the HBM solo launch */
            const uint32_t n_front = db->plan.n_hbm + db->plan.n_hard - n_c; /* predicted-hard records after them */
            uint32_t solo = 0, solo_regions = 0, hbm_solo = 0;
            if (solo_ok && use[1] && n_front &&
                (size_t)ctx->lds2_bytes_per_wave <= 2 * (size_t)waves_per_block * (size_t)ctx->lds_bytes_per_wave) {
                solo = n_front < (uint32_t)ctx->solo_blocks_max ? n_front : (uint32_t)ctx->solo_blocks_max;
                if (solo > blocks / 4) solo = blocks / 4;
                const uint64_t cap = (uint64_t)solo * (uint64_t)ctx->solo_regions_per_wave;
                solo_regions = n_front < cap ? n_front : (uint32_t)cap; /* the others lead the bulk list */
            }
            if (n_c) {
                hbm_solo = (n_c + 3) / 4;
                if (hbm_solo > hbm_solo_max) hbm_solo = hbm_solo_max;
            }
            const int solo_list = launch[1] ? 1 : 0; /* the list the first HBM launch reads */
            const bool later = last > solo_list;
            if (solo || hbm_solo) AVK_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream)); /* after the memsets */
            if (hbm_solo && use_wide && db->plan.n_hbm_notwide && 2u * (db->plan.n_hbm_notwide < n_c ? db->plan.n_hbm_notwide : n_c) <= n_c && !db->notwide_ready) { /* (only the class's wide launch with its companion reads the list) */
                /* the list of the class C records that are not avk_wide.inl's: once per batch, and the FIRST thing queued behind the fork — made behind the class's
                 * wide launch its 20-170 workgroups waited for a place among the step's persistent waves (0.2 ms in a queued shard step, 1.2 ms in a merge call) */
                AVK_HIP(ctx, hipStreamWaitEvent(ctx->wide_stream, ctx->ev_fork, 0));
                AVK_HIP(ctx, hipMemsetAsync(db->d_notwide + n + 1, 0, sizeof(uint32_t), ctx->wide_stream));
                hipLaunchKernelGGL(avk_notwide_list_kernel, dim3((n_c + 255u) / 256u), dim3(256), 0, ctx->wide_stream, db->d_regions, n_c, db->d_notwide, db->d_notwide + n + 1);
                AVK_HIP(ctx, hipGetLastError());
                db->notwide_ready = true;
            }
            bool wide_c = false, wide_x = false;
            if (hbm_solo) {
                AvkKernelArgs s = a;
                AVK_HIP(ctx, hipStreamWaitEvent(ctx->side_stream2, ctx->ev_fork, 0));
                /* how many of them are not the wide kernel's by their own record (a long window, many calls on a side): none or few in a genome, all of them in a
                 * batch of large windows (--min-variant-gap 1000) — which then keeps the launches of round 3: every HBM wave on the whole list */
                const uint32_t n_nw = db->plan.n_hbm_notwide < n_c ? db->plan.n_hbm_notwide : n_c;
                if (use_wide && 2u * n_nw <= n_c) { /* class C through avk_wide.inl first; the HBM launch behind it on this stream takes the list of what that could not take */
                    AvkKernelArgs w = a;
                    w.work_list = nullptr;
                    w.n_work_dev = nullptr;
                    w.work_base = 0;
                    w.n_work = n_c;
                    w.work_counter = db->d_counters + 1240;
                    w.overflow_list = db->d_overflow5;
                    w.overflow_count = db->d_counters + 1244;
                    /* (512 one-wave workgroups for a genome's 6,000-15,000 records; a merge job's 42,000 lasted 7.2 ms on them, the longest launch of its step: one
                     * workgroup per 32 records up to four times as many — profiles/r06_merge_step.txt) */
                    uint32_t wb = (uint32_t)ctx->wide_blocks;
                    if (!getenv("AVK_WIDE_BLOCKS") && n_c / 32u > wb) wb = n_c / 32u < 4u * wb ? n_c / 32u : 4u * wb;
                    uint32_t wg = n_c < wb ? n_c : wb;
                    /* workgroups of the launch for the records that are not the wide kernel's: slices from the END of the solo launch's share, which keeps at least half of
                     * it — a batch with large per-wave slices (adaptive_ws under its budget) may have a share of eight workgroups, and the two launches must never meet on
                     * a slice (they did: xb == hbm_solo_max left the solo launch one workgroup ON the other's first slices — wrong results in a fuzz case where every
                     * region was planned as class C, profiles/r04_gpu_fuzz_classc.txt) */
                    /* (a wave for each of these records and more: they lead the list, most calls first, and a wave that claimed four of them at a time solved four long
                     * searches one after the other — 1.4 ms at the end of a rank's shard whose lanes were done after 0.8, profiles/r05_small_batches.txt) */
                    uint32_t xb = n_nw;
                    xb = xb < hbm_solo ? xb : hbm_solo;
                    xb = xb < 64u ? xb : 64u;
                    if (2u * xb > hbm_solo_max) xb = hbm_solo_max / 2u;
                    avk::wide::WideArgs wc = wa;
                    wc.skip_static = n_nw && xb ? 1u : 0u; /* (without the launch below such a record is handed over like any other) */
                    hipLaunchKernelGGL(avk_wide_kernel, dim3(wg), dim3(64), (size_t)ctx->wide_lds_bytes, ctx->side_stream2, w, wc);
                    AVK_HIP(ctx, hipGetLastError());
                    if (n_nw && xb) { /* the records of class C that are not for the wide kernel by what they say themselves (avk_wide_static_ok: a long window, many calls on a
                       * side) start at the same time, on the HBM-tier kernel and a stream of their own: they are few and each of them is long */
                        AvkKernelArgs x = s;
                        x.pass_tier = 2;
                        x.only_not_wide = 1;
                        x.work_list = nullptr;
                        x.n_work_dev = nullptr;
                        x.work_base = 0;
                        x.n_work = n_c;
                        x.work_counter = db->d_counters + 1256;
                        x.static_pct = 0;
                        x.n_shards = 1;
                        x.n_waves = xb * waves_per_block;
                        /* a wave per record, one claim each, from a list made just ahead of the launch (walking the whole class in claims of four, a wave solved the
                         * records of a claim one after the other: 2.06 -> 1.91 ms for the genome's 27, 0.96 -> 0.90 for a shard's 7) */
                        x.only_not_wide = 0;
                        x.work_list = db->d_notwide;
                        x.n_work_dev = db->d_notwide + n + 1;
                        x.n_work = 0;
                        x.claim = 1;
                        x.high_priority = 1;
                        x.hbm_ws = ctx->d_ws + (size_t)(n_waves + (hbm_solo_max - xb) * waves_per_block) * (size_t)ws_bytes; /* the last slices of the solo launch's share (that launch gets the others) */
                        x.big_ws = ctx->d_big;
                        x.big_busy = db->d_counters + 1088;
                        x.big_slots = big_slots;
                        x.overflow_list = nullptr;
                        x.overflow_count = nullptr;
                        AVK_HIP(ctx, hipStreamWaitEvent(ctx->wide_stream, ctx->ev_fork, 0));
                        /* (a wave per region here: these long windows hold a handful of calls, their searches are short chains where a team's hand-overs cost more
                         * than its parallel pieces give — shard 1.19 -> 1.31 ms, dense mix 2.29 -> 2.53 with teams, profiles/r06_team.txt) */
                        hipLaunchKernelGGL(avk_region_kernel_hbm, dim3(xb), dim3(256), 0, ctx->wide_stream, x);
                        AVK_HIP(ctx, hipGetLastError());
                        AVK_HIP(ctx, hipEventRecord(ctx->ev_wide, ctx->wide_stream));
                        wide_x = true;
                        if (hbm_solo > hbm_solo_max - xb) hbm_solo = hbm_solo_max - xb; /* (at least half of the share: xb <= hbm_solo_max / 2) */
                    }
                    wide_c = true;
                    s.work_list = db->d_overflow5;
                    s.n_work_dev = db->d_counters + 1244;
                    if (ctx->wide_retry_lds_bytes > ctx->wide_lds_bytes) {
                        /* what it hands over at run time is nearly always a search that outgrew the 16 KB (nodes, wavefront blocks): once more with the LDS of a
                         * whole workgroup, a few waves — the wave-per-region kernel needs 2 ms for such a region, at the end of this chain */
                        AvkKernelArgs w2 = w;
                        w2.work_list = db->d_overflow5;
                        w2.n_work_dev = db->d_counters + 1244;
                        w2.work_base = 0;
                        w2.n_work = 0;
                        w2.work_counter = db->d_counters + 1268;
                        w2.overflow_list = db->d_overflow8;
                        w2.overflow_count = db->d_counters + 1272;
                        avk::wide::WideArgs wb = wa;
                        wb.lds_words = (uint32_t)(ctx->wide_retry_lds_bytes / 4);
                        hipLaunchKernelGGL(avk_wide_kernel, dim3(32), dim3(64), (size_t)ctx->wide_retry_lds_bytes, ctx->side_stream2, w2, wb);
                        AVK_HIP(ctx, hipGetLastError());
                        s.work_list = db->d_overflow8;
                        s.n_work_dev = db->d_counters + 1272;
                    }
                }
                uint32_t team_head = 0; /* records at the front of class C (most calls first) that a team launch takes */
                if (!wide_c && ctx->team_long_windows && ctx->team_head_regions > 0 && n_c > 1) {
                    /* a batch of large windows (--min-variant-gap 1000: every region is class C and none is the wide kernel's): its step is as long as its few longest
                     * searches — 0.25 s on one wave for 92 calls on 20 kbp while the other 49,000 regions need 0.05 s of the whole chip — so the head of the class, which
                     * is sorted most calls first, gets a workgroup per region; the other launches start behind it in the list */
                    uint32_t xb = n_c / 2u < (uint32_t)ctx->team_head_regions ? n_c / 2u : (uint32_t)ctx->team_head_regions;
                    if (xb > hbm_solo_max / 2u) xb = hbm_solo_max / 2u;
                    if (xb) {
                        AvkKernelArgs x = s;
                        x.pass_tier = 2;
                        x.only_not_wide = 0;
                        x.team = (uint32_t)ctx->team_long_windows;
                        x.work_list = nullptr;
                        x.n_work_dev = nullptr;
                        x.work_base = 0;
                        x.n_work = xb;
                        x.work_counter = db->d_counters + 1256;
                        x.static_pct = 0;
                        x.n_shards = 1;
                        x.n_waves = xb * waves_per_block;
                        x.claim = 1;
                        x.high_priority = 1;
                        x.hbm_ws = ctx->d_ws + (size_t)(n_waves + (hbm_solo_max - xb) * waves_per_block) * (size_t)ws_bytes; /* the last slices of the solo launch's share */
                        x.big_ws = ctx->d_big;
                        x.big_busy = db->d_counters + 1088;
                        x.big_slots = big_slots;
                        x.overflow_list = nullptr;
                        x.overflow_count = nullptr;
                        AVK_HIP(ctx, hipStreamWaitEvent(ctx->wide_stream, ctx->ev_fork, 0));
                        hipLaunchKernelGGL(avk_region_kernel_team, dim3(xb), dim3(256), 0, ctx->wide_stream, x);
                        AVK_HIP(ctx, hipGetLastError());
                        AVK_HIP(ctx, hipEventRecord(ctx->ev_wide, ctx->wide_stream));
                        wide_x = true;
                        team_head = xb;
                        if (hbm_solo > hbm_solo_max - xb) hbm_solo = hbm_solo_max - xb;
                    }
                }
                s.pass_tier = 2;
                s.work_base = team_head;
                s.n_work = n_c - team_head;
                s.work_counter = db->d_counters + 1076;
                s.static_pct = 0;
                s.n_shards = 1;
                s.claim = 1;
                s.n_waves = hbm_solo * waves_per_block;
                s.high_priority = 1;
                s.hbm_ws = ctx->d_ws + (size_t)n_waves * (size_t)ws_bytes; /* its own slices: the main stream's HBM launch may run beside it */
                s.big_ws = ctx->d_big;
                s.big_busy = db->d_counters + 1088;
                s.big_slots = big_slots;
                s.overflow_list = nullptr; /* same capacities as the last tier: what does not fit fails with CAPACITY */
                s.overflow_count = nullptr;
                if (!big_slots && launch[3]) { /* the big tier has a launch of its own */
                    s.overflow_list = lists[launch[1] ? 2 : 1];
                    s.overflow_count = db->d_counters + 1024 + 16 * (launch[1] ? 2 : 1);
                }
                hipLaunchKernelGGL(avk_region_kernel_hbm, dim3(hbm_solo), dim3(256), 0, ctx->side_stream2, s);
                AVK_HIP(ctx, hipGetLastError());
                AVK_HIP(ctx, hipEventRecord(ctx->ev_join2, ctx->side_stream2));
                hbm_solo_pending = true;
                hbm_shared = !wide_c; /* (the records of class C are the wide launch's: the main stream's HBM launch has nothing to share) */
                hbm_shared_base = team_head;
                wide_x_pending = wide_x;
            }
            if (solo) {
                AvkKernelArgs s = a;
                s.pass_tier = 1;
                s.work_base = n_c;
                s.n_work = solo_regions;
                s.work_counter = db->d_counters + 1072;
                s.static_pct = 0;
                s.n_shards = 1;
                s.claim = 1;
                s.n_waves = solo;
                s.high_priority = 1;
                s.overflow_list = later ? lists[solo_list] : nullptr;
                s.overflow_count = later ? db->d_counters + 1024 + 16 * solo_list : nullptr;
                AVK_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
                hipLaunchKernelGGL(avk_region_kernel_lds, dim3(solo), dim3(64), (size_t)ctx->lds2_bytes_per_wave, ctx->side_stream, s);
                AVK_HIP(ctx, hipGetLastError());
                AVK_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
                solo_pending = true;
            }
            if (solo || hbm_solo) {
                a.work_base = n_c + solo_regions;
                a.n_work = (uint32_t)n - n_fast - n_c - solo_regions;
            }
            /* ---- the lane-per-region launches (avk_lane.inl): the classes with two calls per side (long, latency-bound tiles at low
             * occupancy) on a stream of their own, the one-call classes on the caller's stream ahead of the bulk.  What a lane cannot
             * finish goes to the DEFERRED list, solved after the bulk by an LDS launch of the wave-per-region kernel. */
            /* lane streams: 0 the two-call classes, 1 the one-call classes, 2 the three-call class, 3 the heads of the two-call classes */
            enum { N_LS = 4 };
            hipStream_t lstream[N_LS] = {ctx->lane_stream, ctx->lane_stream2, ctx->lane_stream3, ctx->lane_stream4};
            hipEvent_t ljoin[N_LS] = {ctx->ev_lane_join, ctx->ev_lane_join2, ctx->ev_lane_join3, ctx->ev_lane_join4};
            bool lused[N_LS] = {false, false, false, false}, ljoined[N_LS] = {false, false, false, false}, early_used = false;
            if (use_fast) {
                if (!ctx->lane_attr_set) {
                    AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_lane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_quad_kernel_wide_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    ctx->lane_attr_set = true;
                }
                AvkKernelArgs f = a;
                f.overflow_list = lists[2];
                f.overflow_count = db->d_counters + 1024 + 32;
                AVK_HIP(ctx, hipEventRecord(ctx->ev_lane_fork, ctx->stream));
                /* hand-backs per chain: the launches of a lane stream append to a list of the stream's own, a head launch to one of its own; a list is read by a launch
                 * of avk_wide.inl that follows its writers in stream order (the chain's list: on the chain's stream, no event; a head's: on the fourth lane stream,
                 * behind the head's event, while the rest of the class runs).  One launch for one list behind ALL lane streams started 60-130 us after the last lane
                 * launch (three event waits) and lasted as long as its longest region — a head's hand-back, 160-340 us — at the very end of the step. */
                chains = use_wide && ctx->wide_lane_handbacks && !ctx->lane_head_stream;
                struct HbSeg {
                    uint32_t *list, *count, *cursor;
                };
                uint32_t hb_off = 0;
                int hb_n = 0, hb_heads = 0;
                auto hb_new = [&](uint32_t cap) {
                    HbSeg g = {lists[2] + hb_off, db->d_counters + 1296 + hb_n, db->d_counters + 1328 + hb_n};
                    hb_off += cap;
                    hb_n += 1;
                    return g;
                };
                auto hb_consume = [&](hipStream_t st, const HbSeg &g) {
                    AvkKernelArgs w = a;
                    w.work_list = g.list;
                    w.n_work_dev = g.count;
                    w.work_base = 0;
                    w.n_work = 0;
                    w.work_counter = g.cursor;
                    w.overflow_list = db->d_overflow7;
                    w.overflow_count = db->d_counters + 1264;
                    hipLaunchKernelGGL(avk_wide_kernel_lazy, dim3((uint32_t)ctx->wide_lazy_blocks), dim3(64), (size_t)ctx->wide_lds_bytes, st, w, wa);
                    return hipGetLastError();
                };
                HbSeg chain_seg[2] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
                if (chains) {
                    uint32_t cap[2] = {0, 0};
                    for (int fc = 0; fc < lane_k; ++fc) {
                        const uint32_t mv = fc == AVK_FAST_PAIR ? 1u : AVK_FAST_CLASS[fc].maxv;
                        if (db->fast_tiles[fc] && mv <= 2) cap[mv == 2 ? 0 : 1] += db->fast_tiles[fc] * 64u;
                    }
                    chain_seg[0] = hb_new(cap[0]);
                    chain_seg[1] = hb_new(cap[1]);
                }
                for (int fc = lane_k - 1; fc >= 0; --fc) {
                    if (!db->fast_tiles[fc]) continue;
                    if (fc == AVK_FAST_PAIR && mode == 0 && ctx->lane_pairs) {
                        /* looked up, not searched (avk_pairs.inl): first on the stream of the one-call classes, BESIDE everything else.  (Alone it needs 0.18 ms;
                         * beside the persistent waves of the other launches its workgroups wait for wave slots and it lasts 1.1 ms — in their shadow, which
                         * measured better than 0.18 ms ahead of them: 3.95 against 4.2 ms per whole-genome step.) */
                        /* (AVK_PAIRS_OWN_STREAM=1: on the fourth lane stream with a hand-back list of its own, the one-call chain not waiting for it — a shard's step
                         * 1.04 -> 1.00 ms, a genome's 2.33 -> 2.39: profiles/r06_handback_chains.txt) */
                        const int li = chains && getenv("AVK_PAIRS_OWN_STREAM") ? 3 : 1;
                        if (!lused[li]) {
                            AVK_HIP(ctx, hipStreamWaitEvent(lstream[li], ctx->ev_lane_fork, 0));
                            lused[li] = true;
                        }
                        const int rt = ensure_pair_table(ctx, cfg->max_branch_factor, lstream[li]);
                        if (rt) return rt;
                        avk::pairs::PairArgs pa;
                        pa.recs = db->d_fast + db->fast_word_base[fc];
                        pa.n_tiles = db->fast_tiles[fc];
                        pa.gen_base = db->plan.fast_base[fc];
                        pa.tab = ctx->d_pair_tab;
                        pa.tile_counter = db->d_counters + 1220 + fc;
                        uint32_t pg = (uint32_t)ctx->n_cus * (uint32_t)(ctx->pair_blocks_per_cu > 0 ? ctx->pair_blocks_per_cu : 1);
                        const uint32_t claims = (pa.n_tiles + avk::pairs::PAIR_CLAIM - 1) / avk::pairs::PAIR_CLAIM;
                        if (pg > (claims + 3u) / 4u) pg = (claims + 3u) / 4u;
                        AvkKernelArgs fp = f;
                        HbSeg pseg = chain_seg[1];
                        if (chains && li == 3) pseg = hb_new(pa.n_tiles * 64u);
                        if (chains) fp.overflow_list = pseg.list, fp.overflow_count = pseg.count;
                        hipLaunchKernelGGL(avk_pair_kernel, dim3(pg), dim3(256), 0, lstream[li], fp, pa);
                        AVK_HIP(ctx, hipGetLastError());
                        if (chains && li == 3) AVK_HIP(ctx, hb_consume(lstream[li], pseg));
                        continue;
                    }
                    /* (the looked-up class in merge mode or with the option off: its records are those of a one-call class) */
                    const AvkFastClass &cl = fc == AVK_FAST_PAIR ? AVK_FAST_CLASS[1] : AVK_FAST_CLASS[fc];
                    avk::lane::LaneArgs la;
                    la.recs = db->d_fast + db->fast_word_base[fc];
                    la.rec_words = AVK_FAST_WORDS_OF(cl.maxv);
                    la.n_tiles = db->fast_tiles[fc];
                    la.tile_counter = db->d_counters + 1220 + fc;
                    la.W = cl.W;
                    la.nm = 1u << cl.maxv;
                    la.ed_max = cl.ed_max;
                    la.qcap = cl.qcap;
                    la.gen_base = db->plan.fast_base[fc];
                    la.lanes_log2 = lane_width_log2(ctx, cl.maxv);
                    /* kept node states: where the expensive searches are — the three-call class and the heads of the others (16 records per wave: half a KB of LDS per
                     * slot; in the 64-wide launches of the regions without estimated edits a slot would be 2 KB) */
                    const uint32_t pool_heavy = ctx->lane_pool < 0 ? (la.nm == 2 ? 2u : (la.nm == 4 ? 4u : 6u)) /* lane_pool_default */ : (uint32_t)ctx->lane_pool;
                    la.pool = ctx->lane_pool < 0 ? (cl.maxv > 2 ? pool_heavy : 0u) : (uint32_t)ctx->lane_pool;
                    la.max_nodes = cl.maxv > 2 ? (uint32_t)ctx->lane_node_cap : 250u;
                    la.max_ed_c = (uint32_t)ctx->lane_metrics_ed_cap;
                    uint32_t grid = 0;
                    const size_t lds = lane_launch_geometry(ctx, la, &grid);
                    if (!lds) return fail(ctx, AVK_E_ARG, "lane kernel class %d does not fit the LDS", fc);
                    const int li = cl.maxv == 2 ? 0 : (cl.maxv == 1 ? 1 : 2);
                    if (!lused[li]) {
                        AVK_HIP(ctx, hipStreamWaitEvent(lstream[li], ctx->ev_lane_fork, 0));
                        lused[li] = true;
                    }
                    if (cl.maxv > 2) {
                        /* The three-call class gives up early on large searches (la.max_nodes): those regions are for whole wavefronts.
                         * They get a list of their own and an HBM-tier launch right behind the class on its stream, so that they are
                         * being solved while the other lane classes still run, not after them. */
                        AvkKernelArgs f3 = f;
                        f3.overflow_list = lists[3];
                        f3.overflow_count = db->d_counters + 1104;
                        hipStream_t es = lstream[li]; /* where the launch for the handed-back regions goes */
                        launch_lane_class(ctx, grid, lds, lstream[li], f3, la);
                        AVK_HIP(ctx, hipGetLastError());
                        /* the launch for what ALL lanes hand back waits for the lane launches only, not for the launch behind this class */
                        AVK_HIP(ctx, hipEventRecord(ljoin[li], lstream[li]));
                        ljoined[li] = true;
                        AvkKernelArgs e = a;
                        e.pass_tier = 2;
                        e.work_list = lists[3];
                        e.n_work_dev = db->d_counters + 1104;
                        e.work_base = 0;
                        e.n_work = 0;
                        e.work_counter = db->d_counters + 1120;
                        e.static_pct = 0;
                        e.n_shards = 1;
                        e.claim = 1;
                        e.esc_bytes = 0;
                        e.esc_enabled = 0;
                        e.high_priority = 0;
                        e.extra_counter = nullptr;
                        e.extra_n = 0;
                        e.overflow_list = nullptr;
                        e.overflow_count = nullptr;
                        e.hbm_ws = ctx->d_ws + (size_t)(n_waves + hbm_solo_max * waves_per_block) * (size_t)ws_bytes;
                        e.big_ws = ctx->d_big;
                        e.big_busy = db->d_counters + 1088;
                        e.big_slots = big_slots;
                        uint32_t eb = hbm_blocks < hbm_early_max ? hbm_blocks : hbm_early_max;
                        if (use_wide) { /* large searches on small windows: avk_wide.inl first, the HBM-tier launch takes what is left */
                            AvkKernelArgs w = e;
                            w.work_counter = db->d_counters + 1248;
                            w.overflow_list = db->d_overflow6;
                            w.overflow_count = db->d_counters + 1252;
                            hipLaunchKernelGGL(avk_wide_kernel_lazy, dim3((uint32_t)ctx->wide_lazy_blocks), dim3(64), (size_t)ctx->wide_lds_bytes, es, w, wa);
                            AVK_HIP(ctx, hipGetLastError());
                            e.work_list = db->d_overflow6;
                            e.n_work_dev = db->d_counters + 1252;
                        }
                        e.n_waves = eb * waves_per_block;
                        hipLaunchKernelGGL(avk_region_kernel_hbm_lazy, dim3(eb), dim3(256), 0, es, e);
                        AVK_HIP(ctx, hipGetLastError());
                        AVK_HIP(ctx, hipEventRecord(ctx->ev_lane_early, es)); /* the caller's stream waits for this one at the end */
                        early_used = true;
                        continue;
                    }
                    /* The head of the class — the tiles of regions with estimated edits, which the cost key puts first — in narrow tiles
                     * of its own: lanes that diverge take turns, so 64 expensive regions in one wave take 64 turns; 16 per wave, four times as
                     * many waves.  Same records, same stream, a launch ahead of the rest of the class. */
                    const uint32_t head_tiles = ctx->lane_head_width ? (db->plan.n_fast_heavy[fc] + 63u) / 64u : 0u;
                    if (head_tiles > 0 && head_tiles < la.n_tiles && (uint32_t)ctx->lane_head_width < (1u << la.lanes_log2)) {
                        avk::lane::LaneArgs hd = la;
                        hd.n_tiles = head_tiles;
                        hd.tile_counter = db->d_counters + 1230 + fc;
                        hd.lanes_log2 = head_width_log2(ctx, db->plan.n_fast_heavy[fc]);
                        hd.pool = pool_heavy;
                        uint32_t hgrid = 0;
                        const size_t hlds = lane_launch_geometry(ctx, hd, &hgrid);
                        const int hi = (cl.maxv == 2 && ctx->lane_head_stream) ? 3 : li; /* a long head runs beside the rest of its class */
                        if (!lused[hi]) {
                            AVK_HIP(ctx, hipStreamWaitEvent(lstream[hi], ctx->ev_lane_fork, 0));
                            lused[hi] = true;
                        }
                        AvkKernelArgs fh = f;
                        HbSeg hseg = {nullptr, nullptr, nullptr};
                        const bool staged = chains && hb_heads < 8 && (cl.maxv == 2 || getenv("AVK_STAGE_ALL_HEADS")); /* (the one-call heads are short: their hand-backs wait for the chain's end) */
                        if (staged) {
                            hseg = hb_new(head_tiles * 64u);
                            fh.overflow_list = hseg.list, fh.overflow_count = hseg.count;
                        } else if (chains) {
                            fh.overflow_list = chain_seg[li].list, fh.overflow_count = chain_seg[li].count;
                        }
                        launch_lane_class(ctx, hgrid, hlds, lstream[hi], fh, hd, staged ? ctx->ev_hb[hb_heads] : nullptr);
                        AVK_HIP(ctx, hipGetLastError());
                        if (staged) { /* its hand-backs: solved beside the rest of the class */
                            AVK_HIP(ctx, hipStreamWaitEvent(lstream[3], ctx->ev_hb[hb_heads], 0));
                            lused[3] = true; /* (its first command waits for an event behind ev_lane_fork) */
                            AVK_HIP(ctx, hb_consume(lstream[3], hseg));
                            hb_heads += 1;
                        }
                        la.recs += (size_t)head_tiles * la.rec_words * 64u;
                        la.n_tiles -= head_tiles;
                        la.gen_base += head_tiles * 64u;
                        if (grid > la.n_tiles) grid = la.n_tiles;
                    }
                    AvkKernelArgs fr = f;
                    if (chains) fr.overflow_list = chain_seg[li].list, fr.overflow_count = chain_seg[li].count;
                    launch_lane_class(ctx, grid, lds, lstream[li], fr, la);
                    AVK_HIP(ctx, hipGetLastError());
                }
                if (chains)
                    for (int li = 0; li < 2; ++li)
                        if (lused[li]) AVK_HIP(ctx, hb_consume(lstream[li], chain_seg[li]));
                for (int li = 0; li < N_LS; ++li) {
                    if (!lused[li]) continue;
                    if (!ljoined[li]) AVK_HIP(ctx, hipEventRecord(ljoin[li], lstream[li]));
                }
                if (lused[1] && timed) { /* end of the one-call classes' launches */
                    AVK_HIP(ctx, hipEventRecord(ctx->ev_lane, ctx->lane_stream2));
                    ctx->ev_lane_valid = true;
                }
            }
            /* the three launches are sized so that all their workgroups CAN be resident at once; with lane launches beside them they are not (a 40 KB LDS
             * allocation waits for a contiguous hole) — nothing waits for a workgroup to start, late ones find the claimed part of the list empty */
            uint32_t bulk = blocks > solo + 2 * hbm_solo + blocks / 4 ? blocks - solo - 2 * hbm_solo : blocks / 4; /* (a large class C launch: the bulk keeps a quarter of its workgroups) */
            if (bulk < 1) bulk = 1;
            if (bulk > blocks) bulk = blocks;
            if (ctx->bulk_full_grid) bulk = blocks;
            const uint32_t bulk_unfit = bulk; /* (the launch for the lanes' hand-backs below is sized by this) */
            if (ctx->bulk_fit && !ctx->bulk_full_grid) {
                const uint32_t fit = (a.n_work + waves_per_block - 1) / waves_per_block;
                if (bulk > fit) bulk = fit < 1 ? 1 : fit;
            }
            a.n_waves = bulk * waves_per_block;
            const uint64_t slice0 = a.tier[0].ws_bytes;
            if (ctx->lds_bytes_per_wave >= 1024) { /* slices shrink to make room for the workgroup's tail */
                a.tier[0].ws_bytes = avk::bulk_slice_bytes((uint64_t)ctx->lds_bytes_per_wave);
                a.esc_bytes = (uint32_t)(waves_per_block * a.tier[0].ws_bytes);
                a.esc_enabled = ctx->lds_escalation ? 1u : 0u;
            }
            hipLaunchKernelGGL(avk_region_kernel_lds, dim3(bulk), dim3(256), (size_t)waves_per_block * (size_t)ctx->lds_bytes_per_wave, ctx->stream, a);
            if (use_fast) {
                /* The regions the lanes handed over: small ones, so the LDS tier with its in-workgroup escalation — on the one-call lane
                 * stream, behind the lane launches, BESIDE the bulk and the HBM launch of this stream.  What overflows there (rare) goes to a
                 * list of its own, read by one more HBM launch at the very end (normally empty: 10 us). */
                AVK_HIP(ctx, hipGetLastError());
                const int di = lused[1] ? 1 : (lused[0] ? 0 : 2);
                hipStream_t ds = lstream[di];
                if (chains) { /* every list has been through avk_wide.inl on its own stream; what that left is for the caller's stream at the end of the step (below) */
                    for (int li = 0; li < N_LS; ++li)
                        if (lused[li]) chain_join[n_chain_join++] = ljoin[li];
                } else {
                    for (int li = 0; li < N_LS; ++li)
                        if (li != di && lused[li]) AVK_HIP(ctx, hipStreamWaitEvent(ds, ljoin[li], 0));
                }
                AvkKernelArgs d = a;
                d.work_list = chains ? db->d_overflow7 : lists[2];
                d.n_work_dev = chains ? db->d_counters + 1264 : db->d_counters + 1024 + 32;
                d.work_base = 0;
                d.n_work = 0;
                d.work_counter = db->d_counters + 768;
                if (!chains && use_wide && ctx->wide_lane_handbacks) { /* small windows, whatever made the lanes give up: avk_wide.inl first, the LDS launch takes what is left */
                    AvkKernelArgs w = d;
                    w.work_counter = db->d_counters + 1260;
                    w.overflow_list = db->d_overflow7;
                    w.overflow_count = db->d_counters + 1264;
                    hipLaunchKernelGGL(avk_wide_kernel_lazy, dim3((uint32_t)ctx->wide_lazy_blocks), dim3(64), (size_t)ctx->wide_lds_bytes, ds, w, wa);
                    AVK_HIP(ctx, hipGetLastError());
                    d.work_list = db->d_overflow7;
                    d.n_work_dev = db->d_counters + 1264;
                }
                uint32_t dblocks = bulk_unfit < (uint32_t)ctx->n_cus ? bulk_unfit : (uint32_t)ctx->n_cus; /* one workgroup per CU: the list is short */
                d.n_waves = dblocks * waves_per_block;
                d.overflow_list = lists[1];
                d.overflow_count = db->d_counters + 1024 + 16;
                if (chains) {
                    chain_d = d;
                    chain_dblocks = dblocks;
                } else {
                    hipLaunchKernelGGL(avk_region_kernel_lds_lazy, dim3(dblocks), dim3(256), (size_t)waves_per_block * (size_t)ctx->lds_bytes_per_wave, ds, d);
                    AVK_HIP(ctx, hipGetLastError());
                    AVK_HIP(ctx, hipEventRecord(ctx->ev_lane_done, ds)); /* everything of the lane streams is behind this record */
                }
                deferred_pending = true;
                early_pending = early_used;
            }
            a.tier[0].ws_bytes = slice0;
        } else if (t == 1) { /* one workgroup per CU, four large slices */
            a.hbm_ws = nullptr;
            uint32_t b2 = (uint32_t)ctx->n_cus < blocks ? (uint32_t)ctx->n_cus : blocks;
            a.n_waves = b2 * waves_per_block;
            hipLaunchKernelGGL(avk_region_kernel_lds, dim3(b2), dim3(256), (size_t)waves_per_block * (size_t)ctx->lds2_bytes_per_wave, ctx->stream, a);
        } else if (t == 2) {
            a.hbm_ws = ctx->d_ws;
            a.big_ws = ctx->d_big;
            a.big_busy = db->d_counters + 1088;
            a.big_slots = big_slots;
            a.n_waves = hbm_blocks * waves_per_block;
            if (hbm_solo_pending && hbm_shared) { /* class C records the solo launch has not started yet: every wave of this launch helps (same ticket counter) */
                a.extra_counter = db->d_counters + 1076;
                a.extra_base = hbm_shared_base;
                a.extra_n = db->plan.n_hbm - hbm_shared_base;
            }
            hipLaunchKernelGGL(avk_region_kernel_hbm, dim3(hbm_blocks), dim3(256), 0, ctx->stream, a);
        } else {
            a.hbm_ws = ctx->d_big;
            a.big_slots = 0;
            a.n_waves = big_blocks * waves_per_block;
            hipLaunchKernelGGL(avk_region_kernel_hbm, dim3(big_blocks), dim3(256), 0, ctx->stream, a);
        }
        AVK_HIP(ctx, hipGetLastError());
        a.extra_counter = nullptr;
        a.extra_n = 0;
        if (first_launch && timed) AVK_HIP(ctx, hipEventRecord(ctx->evk1, ctx->stream));

        if (t != last) {
            list = lists[nlist];
            count = db->d_counters + 1024 + 16 * nlist;
            nlist += 1;
        }
    }
    if (solo_pending) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    if (hbm_solo_pending) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join2, 0));
    if (wide_x_pending) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_wide, 0));
    if (deferred_pending) { /* the lane streams (lane launches, then the handed-back regions) join here; what even the escalation could not hold */
        if (!chains) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lane_done, 0));
        if (early_pending) AVK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lane_early, 0));
        AvkKernelArgs h = a;
        h.pass_tier = 2;
        h.work_list = lists[1];
        h.n_work_dev = db->d_counters + 1024 + 16;
        h.work_base = 0;
        h.n_work = 0;
        h.work_counter = db->d_counters + 256; /* the claim counters of the (unused) tier-1 launch */
        h.static_pct = 0;
        h.n_shards = 1;
        h.claim = 1;
        h.esc_bytes = 0;
        h.esc_enabled = 0;
        h.high_priority = 0;
        h.extra_counter = nullptr;
        h.extra_n = 0;
        h.overflow_list = nullptr;
        h.overflow_count = nullptr;
        h.hbm_ws = ctx->d_ws;
        h.big_ws = ctx->d_big;
        h.big_busy = db->d_counters + 1088;
        h.big_slots = big_slots;
        const uint32_t hb = hbm_blocks < 64 ? hbm_blocks : 64;
        h.n_waves = hb * waves_per_block;
        hipLaunchKernelGGL(avk_region_kernel_hbm_lazy, dim3(hb), dim3(256), 0, ctx->stream, h);
        AVK_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(avk_tally_reduce, dim3((AVK_TALLY_STRIDE + 63) / 64), dim3(64), 0, ctx->stream, db->d_partials, db->d_tally,
                       (uint64_t *)tally_dev, db->d_counters, (unsigned)AVK_N_COUNTERS, ctx->accumulate_tally ? 1u : 0u);
    AVK_HIP(ctx, hipGetLastError());
    db->scratch_clean = true;
    db->last_cfg = *cfg;
    db->last_mode = mode;
    db->has_run = true;
    if (timed) AVK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = timed;
    return 0;
}

int avk_compare_resident(avk_ctx *ctx, avk_dev_batch *db, const avk_compare_config *cfg, void *tally_dev) {
    return run_internal(ctx, db, cfg, tally_dev, 0);
}

#ifdef AVK_QUAD_WAVE_LOG
/* profiling build only (make quad-wave-log): the wave log of the quad launches, 6 x 4096 x 4 words; cleared by the call */
extern "C" int avk_debug_wave_log(unsigned long long *dst) {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(avk_wave_log_buf), sizeof(avk_wave_log_buf)) != hipSuccess) return AVK_E_HIP;
    std::vector<unsigned long long> z(sizeof(avk_wave_log_buf) / 8, 0);
    return hipMemcpyToSymbol(HIP_SYMBOL(avk_wave_log_buf), z.data(), sizeof(avk_wave_log_buf)) == hipSuccess ? 0 : AVK_E_HIP;
}
#endif

int avk_synchronize(avk_ctx *ctx) {
    if (!ctx) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int avk_last_kernel_ms(avk_ctx *ctx, float *ms) {
    if (!ctx || !ms) return AVK_E_ARG;
    if (!ctx->ev_valid) return fail(ctx, AVK_E_STATE, "no launch has been timed yet");
    AVK_HIP(ctx, hipEventSynchronize(ctx->evk1));
    AVK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->evk1));
    return 0;
}

int avk_last_lane_ms(avk_ctx *ctx, float *ms) {
    if (!ctx || !ms) return AVK_E_ARG;
    *ms = 0;
    if (!ctx->ev_valid) return fail(ctx, AVK_E_STATE, "no launch has been timed yet");
    if (!ctx->ev_lane_valid) return 0; /* the last call had no lane-kernel launches */
    AVK_HIP(ctx, hipEventSynchronize(ctx->ev_lane));
    AVK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev_lane));
    return 0;
}

int avk_last_lane_solved(avk_ctx *ctx, uint64_t *count) {
    if (!ctx || !count) return AVK_E_ARG;
    *count = ctx->last_lane_solved;
    return 0;
}

int avk_last_wide_solved(avk_ctx *ctx, uint64_t *count) {
    if (!ctx || !count) return AVK_E_ARG;
    *count = ctx->last_wide_solved;
    return 0;
}

int avk_last_solver_ms(avk_ctx *ctx, float *ms) {
    if (!ctx || !ms) return AVK_E_ARG;
    if (!ctx->ev_valid) return fail(ctx, AVK_E_STATE, "no launch has been timed yet");
    AVK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    AVK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return 0;
}

/* profiling builds (-DAVK_PHASE_TIMING): summed s_memtime ticks per solver phase of the last download */
int avk_debug_phase_cycles(avk_ctx *ctx, uint64_t out[16]) {
    if (!ctx || !out) return AVK_E_ARG;
    memcpy(out, ctx->last_phase, sizeof(ctx->last_phase));
    return 0;
}

/* diagnostic: which of the context's streams still have work queued (caller's, two solo streams, two lane streams) and the batch's
 * device counters as they are NOW (copied on a stream of its own, beside whatever is running) */
int avk_debug_snapshot(avk_ctx *ctx, avk_dev_batch *db, uint32_t *counters, uint32_t n_counters, int32_t busy[5]) {
    if (!ctx || !db || !counters || !busy) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    static hipStream_t probe = nullptr;
    if (!probe) AVK_HIP(ctx, hipStreamCreateWithFlags(&probe, hipStreamNonBlocking));
    hipStream_t ss[5] = {ctx->stream, ctx->side_stream, ctx->side_stream2, ctx->lane_stream, ctx->lane_stream2}; /* (lane_stream3 joins lane_stream2 or lane_stream) */
    for (int i = 0; i < 5; ++i) busy[i] = ss[i] ? (hipStreamQuery(ss[i]) == hipErrorNotReady ? 1 : 0) : -1;
    (void)hipGetLastError();
    if (n_counters > AVK_N_COUNTERS) n_counters = AVK_N_COUNTERS;
    AVK_HIP(ctx, hipMemcpyAsync(counters, db->d_counters, (size_t)n_counters * sizeof(uint32_t), hipMemcpyDeviceToHost, probe));
    AVK_HIP(ctx, hipStreamSynchronize(probe));
    return 0;
}

/* The work order of a device-packed batch and the plan behind it: region order[k] is record k; [0, n_hbm) class C, [n_hbm, n_hbm + n_hard) class B, then the bulk, then the
 * lane classes from fast_base[fc] (n_fast[fc] regions each, the first n_fast_heavy[fc] its head); class AVK_FAST_PAIR is looked up.  A measurement aid: bench.py prices every
 * launch class with the algorithmic bytes of ITS regions.  counts[0..3] = n_hbm, n_hbm_notwide, n_hard, n_fast_total; [4 + 3 fc ..] = fast_base, n_fast, n_fast_heavy of
 * class fc (22 words).  order may be NULL.  Host-packed batches: AVK_E_STATE. */
int avk_debug_work_order(avk_ctx *ctx, avk_dev_batch *db, uint32_t *order, uint64_t counts[22]) {
    if (!ctx || !db || !counts) return AVK_E_ARG;
    if (!db->dev_packed || !db->dp_args.order) return fail(ctx, AVK_E_STATE, "the batch was not packed on the device");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    counts[0] = db->plan.n_hbm, counts[1] = db->plan.n_hbm_notwide, counts[2] = db->plan.n_hard, counts[3] = db->plan.n_fast_total;
    for (int fc = 0; fc < AVK_FAST_CLASSES; ++fc) counts[4 + 3 * fc] = db->plan.fast_base[fc], counts[5 + 3 * fc] = db->plan.n_fast[fc], counts[6 + 3 * fc] = db->plan.n_fast_heavy[fc];
    if (order && db->n_regions) {
        AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AVK_HIP(ctx, hipMemcpy(order, db->dp_args.order, (size_t)db->n_regions * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int avk_last_tier_counts(avk_ctx *ctx, uint64_t counts[5]) {
    if (!ctx || !counts) return AVK_E_ARG;
    memcpy(counts, ctx->last_tiers, sizeof(ctx->last_tiers));
    return 0;
}


/* Regions that exhausted the last workspace tier (AVK_ST_CAPACITY) are solved again by the library itself, in slices of 1, 4 and 16 GB
 * (the reference grows its vectors without bound, dynamic_wfa.rs:152; an MI355X has 288 GB): the regions are rebuilt from the packed
 * batch the context keeps, sent through the big-slice tier alone, and their results are written over the failed ones.  `fixed`
 * receives, per level, the sub-batch results. */
namespace {
struct CapacityFix {
    std::vector<uint32_t> idx; /* caller region index of the sub-batch's regions */
    std::vector<int32_t> status;
    std::vector<uint32_t> ed1, ed2, nopt, seq_len, gm, bp_off, bp;
    std::vector<uint16_t> present;
    std::vector<uint8_t> ve, vo, vc, vz, seq;
    std::vector<uint64_t> seq_off, v_first; /* first sub-batch variant of a region (truth first, then query) */
    std::vector<uint32_t> seq_stride;
    std::vector<uint64_t> tally;
};
} // namespace

static int rerun_capacity_regions(avk_ctx *ctx, avk_dev_batch *db, const std::vector<uint32_t> &cap, bool want_gm, bool want_seq, bool want_bp, uint64_t slice_bytes, CapacityFix *fx) {
    const uint64_t m = cap.size();
    std::vector<uint64_t> rid(m), st(m), en(m), toff(m), qoff(m), vpos, a0off, a1off;
    std::vector<uint32_t> cidx(m), tcnt(m), qcnt(m), vraw, a0len, a1len;
    std::vector<uint8_t> vtype, vzyg, arena;
    fx->idx = cap;
    fx->v_first.assign(m, 0);
    for (uint64_t k = 0; k < m; ++k) {
        const AvkDevRegion &dr = db->host.regions[cap[k]];
        uint32_t c = 0;
        while (c + 1 < ctx->contig_base.size() && ctx->contig_base[c + 1] <= dr.ref_off) c += 1;
        rid[k] = cap[k];
        cidx[k] = c;
        st[k] = dr.ref_off - ctx->contig_base[c];
        en[k] = st[k] + dr.len;
        const uint32_t N = dr.t_cnt + dr.q_cnt;
        const uint8_t *base = (const uint8_t *)(db->host.blob.data() + 2ull * dr.blob_off);
        const AvkBlobVar *bv = (const AvkBlobVar *)base;
        const uint8_t *ba = base + (((uint64_t)N * sizeof(AvkBlobVar) + 15) & ~15ull);
        toff[k] = vpos.size();
        tcnt[k] = dr.t_cnt;
        qoff[k] = vpos.size() + dr.t_cnt;
        qcnt[k] = dr.q_cnt;
        fx->v_first[k] = vpos.size();
        for (uint32_t i = 0; i < N; ++i) {
            vpos.push_back(st[k] + bv[i].rel_pos);
            vtype.push_back((uint8_t)(bv[i].type_zyg & 0xFF));
            vzyg.push_back((uint8_t)((bv[i].type_zyg >> 8) & 0xFF));
            vraw.push_back(bv[i].raw_space);
            a0off.push_back(arena.size());
            a0len.push_back(bv[i].a0_len);
            arena.insert(arena.end(), ba + bv[i].a_off, ba + bv[i].a_off + bv[i].a0_len);
            a1off.push_back(arena.size());
            a1len.push_back(bv[i].a1_len);
            arena.insert(arena.end(), ba + bv[i].a_off + bv[i].a0_len, ba + bv[i].a_off + bv[i].a0_len + bv[i].a1_len);
        }
    }
    arena.push_back(0);
    avk_region_batch b;
    memset(&b, 0, sizeof(b));
    b.n_regions = m;
    b.region_id = rid.data(), b.contig_idx = cidx.data(), b.start = st.data(), b.end = en.data(), b.t_off = toff.data(), b.t_cnt = tcnt.data(), b.q_off = qoff.data(),
    b.q_cnt = qcnt.data();
    b.n_variants = vpos.size();
    b.var_pos = vpos.data(), b.var_type = vtype.data(), b.var_zyg = vzyg.data(), b.var_raw_space = vraw.data(), b.a0_off = a0off.data(), b.a0_len = a0len.data(),
    b.a1_off = a1off.data(), b.a1_len = a1len.data(), b.allele_bytes = arena.data(), b.allele_bytes_len = arena.size();
    const uint64_t nvs = vpos.size();
    fx->status.assign(m, 0), fx->ed1.assign(m, 0), fx->ed2.assign(m, 0), fx->nopt.assign(m, 0), fx->present.assign(m, 0);
    fx->ve.assign(nvs + 1, 0), fx->vo.assign(nvs + 1, 0), fx->vc.assign(nvs + 1, 0), fx->vz.assign(nvs + 1, 0);
    fx->tally.assign(AVK_TALLY_LEN, 0);
    avk_result_batch o;
    memset(&o, 0, sizeof(o));
    o.status = fx->status.data(), o.ed_h1 = fx->ed1.data(), o.ed_h2 = fx->ed2.data(), o.n_optima = fx->nopt.data(), o.type_present = fx->present.data();
    o.var_expected = fx->ve.data(), o.var_observed = fx->vo.data(), o.var_class = fx->vc.data(), o.var_zyg = fx->vz.data(), o.tally = fx->tally.data();
    if (want_gm) {
        fx->gm.assign(m * AVK_N_GROUPS * AVK_N_FIELDS, 0);
        o.group_metrics = fx->gm.data();
    }
    const int64_t keep_bp = ctx->emit_bp_groups;
    if (want_bp) {
        fx->bp_off.assign(m + 1, 0);
        fx->bp.assign((m + nvs + 1) * 4, 0);
        o.bp_off = fx->bp_off.data(), o.bp_groups = fx->bp.data();
    }
    ctx->emit_bp_groups = want_bp ? 1 : 0;
    if (want_seq) {
        fx->seq_off.assign(m, 0), fx->seq_stride.assign(m, 0), fx->seq_len.assign(m * 5, 0);
        uint64_t total = 0;
        for (uint64_t k = 0; k < m; ++k) {
            fx->seq_stride[k] = avk::seq_stride_of(&b, k);
            fx->seq_off[k] = total;
            total += 5ull * fx->seq_stride[k];
        }
        fx->seq.assign(total + 16, 0);
        o.seq_bytes = fx->seq.data(), o.seq_off = fx->seq_off.data(), o.seq_stride = fx->seq_stride.data(), o.seq_len = fx->seq_len.data();
    }
    /* the big-slice tier alone */
    const int64_t keep[] = {ctx->lds_bytes_per_wave, ctx->lds2_bytes_per_wave, ctx->ws_bytes_per_wave, ctx->big_ws_bytes, ctx->big_waves, ctx->lane_kernel, ctx->capacity_retry,
                            ctx->emit_group_metrics};
    ctx->lds_bytes_per_wave = 0, ctx->lds2_bytes_per_wave = 0, ctx->ws_bytes_per_wave = 0, ctx->big_ws_bytes = (int64_t)slice_bytes, ctx->big_waves = m < 4 ? (int64_t)m : 4,
    ctx->lane_kernel = 0, ctx->capacity_retry = 0, ctx->emit_group_metrics = want_gm ? 1 : 0;
    avk_dev_batch *sub = nullptr;
    int rc = upload_internal(ctx, &b, false, &sub);
    if (!rc) rc = run_internal(ctx, sub, &db->last_cfg, nullptr, 0);
    if (!rc) rc = avk_results_download(ctx, sub, &o);
    if (sub) avk_batch_free(ctx, sub);
    ctx->lds_bytes_per_wave = keep[0], ctx->lds2_bytes_per_wave = keep[1], ctx->ws_bytes_per_wave = keep[2], ctx->big_ws_bytes = keep[3], ctx->big_waves = keep[4],
    ctx->lane_kernel = keep[5], ctx->capacity_retry = keep[6], ctx->emit_group_metrics = keep[7];
    ctx->emit_bp_groups = keep_bp;
    return rc;
}

} /* extern "C" */
/* avk_results_download in one piece (later == NULL), or in the two pieces of the asynchronous boundary: everything up to the queued copies (later given, tally_ready
 * NULL), and everything behind them — the tally, the statistics, the capacity retry — once the copies have arrived (tally_ready = the batch's tally words) */
static int results_download_impl(avk_ctx *ctx, avk_dev_batch *db, avk_result_batch *out, DownloadLater *later, const uint64_t *tally_ready) {
    if (!ctx || !db || !out || !(out->status || out->region_packed)) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = db->n_regions, nv = db->n_variants_dev;
    const bool timing = getenv("AVK_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<uint64_t> tally((size_t)AVK_TALLY_STRIDE);
    if (tally_ready) memcpy(tally.data(), tally_ready, (size_t)AVK_TALLY_STRIDE * sizeof(uint64_t));
    if (later && !(db->dev_packed && db->var_dense && !(out->seq_bytes && out->seq_len && out->seq_off && out->seq_stride && db->d_seq)))
        return fail(ctx, AVK_E_STATE, "results can only be queued for a device-packed batch that owns all its calls, without sequence outputs");
    hipStream_t s = ctx->stream;
    const bool want_seq = out->seq_bytes && out->seq_len && out->seq_off && out->seq_stride && db->d_seq;
    std::vector<uint8_t> seq;
    std::vector<uint32_t> seqlen;
#define D2H(dst, src, bytes) \
    if ((bytes) > 0) AVK_HIP(ctx, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, s))
    if (want_seq) {
        seq.resize(db->seq_total + 16);
        seqlen.resize(n * 5 + 1);
        D2H(seq.data(), db->d_seq, db->seq_total);
        D2H(seqlen.data(), db->d_seqlen, n * 5 * sizeof(uint32_t));
    }
    std::chrono::steady_clock::time_point t_copied;
    /* the host-side writers of a region's and a call's results: whichever of the wide arrays and of the packed form the caller handed in */
    auto put_region = [&](uint64_t r, uint32_t st, uint32_t e1, uint32_t e2, uint32_t nopt, uint32_t present) {
        if (out->status) out->status[r] = (int32_t)st;
        if (out->ed_h1) out->ed_h1[r] = e1;
        if (out->ed_h2) out->ed_h2[r] = e2;
        if (out->n_optima) out->n_optima[r] = nopt;
        if (out->type_present) out->type_present[r] = (uint16_t)present;
        if (out->region_packed) out->region_packed[r] = avk_rp_make(st, e1, e2, nopt, present);
    };
    auto put_call = [&](uint64_t hv, uint32_t ea, uint32_t oa, uint32_t cls, uint32_t zyg) {
        if (out->var_expected) out->var_expected[hv] = (uint8_t)ea;
        if (out->var_observed) out->var_observed[hv] = (uint8_t)oa;
        if (out->var_class) out->var_class[hv] = (uint8_t)cls;
        if (out->var_zyg) out->var_zyg[hv] = (uint8_t)zyg;
        if (out->var_packed) out->var_packed[hv] = avk_vp_make(ea, oa, zyg);
    };
    auto status_of = [&](uint64_t r) -> int32_t { return out->status ? out->status[r] : avk_rp_status(out->region_packed[r]); };
    if (tally_ready) { /* the copies were queued earlier and have arrived */
        t_copied = std::chrono::steady_clock::now();
    } else if (db->dev_packed) { /* the caller's layout is made on the device (dp_unpack); the copies land in the caller's arrays */
        const int rc = download_device_packed(ctx, db, out, nullptr, tally.data(), later);
        if (rc || later) return rc;
        t_copied = std::chrono::steady_clock::now();
        if (want_seq) {
            const int rv = materialize_host_view(ctx, db);
            if (rv) return rv;
        }
    } else {
        std::vector<uint32_t> rout(n * 4 + 4), vout(nv + 1);
        D2H(rout.data(), db->d_region_out, n * 4 * sizeof(uint32_t));
        if (out->group_metrics && ctx->emit_group_metrics && db->d_gm) D2H(out->group_metrics, db->d_gm, n * AVK_N_GROUPS * AVK_N_FIELDS * sizeof(uint32_t));
        if (out->bp_off && out->bp_groups && db->d_bp && db->d_bp_off) {
            D2H(out->bp_off, db->d_bp_off, (n + 1) * sizeof(uint32_t));
            D2H(out->bp_groups, db->d_bp, (size_t)db->n_bp_groups * 4 * sizeof(uint32_t));
        }
        const bool want_calls = out->var_expected || out->var_observed || out->var_class || out->var_zyg || out->var_packed;
        if (want_calls) D2H(vout.data(), db->d_var_out, nv * sizeof(uint32_t));
        D2H(tally.data(), db->d_tally, (size_t)AVK_TALLY_STRIDE * sizeof(uint64_t));
        AVK_HIP(ctx, hipStreamSynchronize(s));
        t_copied = std::chrono::steady_clock::now();
        for (uint64_t r = 0; r < n; ++r) {
            const uint32_t *w = rout.data() + 4 * r;
            put_region(r, w[0], w[1], w[2], w[3] & 0xFFFFu, w[3] >> 16);
        }
        for (uint64_t v = 0; want_calls && v < nv; ++v) {
            const uint32_t w = vout[v];
            put_call(db->host.dev2host[v], w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF, w >> 24);
        }
    }
#undef D2H
    if (timing)
        fprintf(stderr, "avk download%s: kernels + copies %.3f ms, unpack %.3f ms\n", db->dev_packed ? " (device-packed)" : "", std::chrono::duration<double, std::milli>(t_copied - t_begin).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_copied).count());
    if (out->tally) memcpy(out->tally, tally.data(), AVK_TALLY_LEN * sizeof(uint64_t));
    memcpy(ctx->last_tiers, tally.data() + AVK_TALLY_LEN, 5 * sizeof(uint64_t));
    ctx->last_lane_solved = tally[AVK_TALLY_LANE_SOLVED];
    ctx->last_wide_solved = tally[AVK_TALLY_WIDE_SOLVED];
    memcpy(ctx->last_phase, tally.data() + AVK_TALLY_LEN + 5, 16 * sizeof(uint64_t));
    if (want_seq) {
        for (uint64_t r = 0; r < n; ++r) {
            const AvkDevRegion &dr = db->host.regions[r];
            for (int k = 0; k < 5; ++k) {
                uint32_t len = seqlen[5 * r + k];
                if (len > out->seq_stride[r]) len = out->seq_stride[r];
                memcpy(out->seq_bytes + out->seq_off[r] + (uint64_t)k * out->seq_stride[r], seq.data() + dr.seq_off + (uint64_t)k * dr.seq_stride, len);
                out->seq_len[5 * r + k] = len;
            }
        }
    }
    /* no solvable region stays a capacity failure: larger slices, level by level (an overflow status is only allowed when
     * the library itself solves the region again) */
    if (ctx->capacity_retry && db->has_run && db->last_mode == 0 && ctx->last_tiers[4] > 0) { /* (the kernels count the regions they give up on: a step without
                                                                                                  any — every step of a genome — does not walk over its 3.6 million
                                                                                                  statuses, 1.4 ms of one host thread) */
        std::vector<uint32_t> cap;
        for (uint64_t r = 0; r < n; ++r)
            if (status_of(r) == AVK_ST_CAPACITY) cap.push_back((uint32_t)r);
        if (!cap.empty()) { /* the retry rebuilds the regions from the packed records: a device-packed batch fetches them now */
            const int rv = materialize_host_view(ctx, db);
            if (rv) return rv;
        }
        const bool big_grown = !cap.empty();
        const size_t big_before = ctx->big_alloc;
        int retry_rc = 0;
        uint64_t keep_tiers[5];
        memcpy(keep_tiers, ctx->last_tiers, sizeof(keep_tiers));
        const uint64_t keep_lane_solved = ctx->last_lane_solved, keep_wide_solved = ctx->last_wide_solved;
        for (uint64_t slice = 1ull << 30; !cap.empty() && slice <= (16ull << 30); slice <<= 2) {
            if ((int64_t)slice <= ctx->big_ws_bytes) continue;
            CapacityFix fx;
            const bool dev_gm = ctx->emit_group_metrics && db->d_gm; /* the batch keeps per-region blocks on the device (avk_label_tallies reads them) */
            const bool want_gm = (out->group_metrics && ctx->emit_group_metrics && db->d_gm) || dev_gm;
            const bool want_bp_words = out->bp_packed && out->bp_spilled && out->bp_groups && db->d_bp;
            const bool want_bp = (out->bp_off && out->bp_groups && db->d_bp) || want_bp_words;
            const int rc = rerun_capacity_regions(ctx, db, cap, want_gm, want_seq, want_bp, slice, &fx);
            if (rc == AVK_E_OOM) break; /* the device cannot hold slices of this size: the regions keep their status */
            if (rc) { /* reported after the statistics and the shared slices are back as they were (below) */
                retry_rc = rc;
                break;
            }
            std::vector<uint32_t> still;
            for (uint64_t k = 0; k < fx.idx.size(); ++k) {
                const uint32_t r = fx.idx[k];
                if (fx.status[k] == AVK_ST_CAPACITY) {
                    still.push_back(r);
                    continue;
                }
                put_region(r, (uint32_t)fx.status[k], fx.ed1[k], fx.ed2[k], fx.nopt[k], fx.present[k]);
                { /* the batch on the device learns of the repair as well: its region record, and its metric block when it keeps them — the per-label sums of
                   * avk_label_tallies count every solved region, as the totals do */
                    const uint32_t w4[4] = {(uint32_t)fx.status[k], fx.ed1[k], fx.ed2[k], fx.nopt[k] | ((uint32_t)fx.present[k] << 16)};
                    hipError_t ew = hipMemcpyAsync(db->d_region_out + 4 * (size_t)r, w4, sizeof(w4), hipMemcpyHostToDevice, s);
                    if (ew == hipSuccess && dev_gm)
                        ew = hipMemcpyAsync(db->d_gm + (size_t)r * AVK_N_GROUPS * AVK_N_FIELDS, fx.gm.data() + (size_t)k * AVK_N_GROUPS * AVK_N_FIELDS,
                                            sizeof(uint32_t) * AVK_N_GROUPS * AVK_N_FIELDS, hipMemcpyHostToDevice, s);
                    if (ew == hipSuccess) ew = hipStreamSynchronize(s); /* (w4 and fx leave scope) */
                    if (ew != hipSuccess && !retry_rc) retry_rc = fail(ctx, AVK_E_HIP, "capacity retry: %s", hipGetErrorString(ew));
                }
                const AvkDevRegion &dr = db->host.regions[r];
                for (uint32_t i = 0; i < dr.t_cnt + dr.q_cnt; ++i) {
                    const uint64_t sv = fx.v_first[k] + i;
                    put_call(db->host.dev2host[dr.v_off + i], fx.ve[sv], fx.vo[sv], fx.vc[sv], fx.vz[sv]);
                }
                if (want_gm && out->group_metrics) memcpy(out->group_metrics + (size_t)r * AVK_N_GROUPS * AVK_N_FIELDS, fx.gm.data() + (size_t)k * AVK_N_GROUPS * AVK_N_FIELDS, sizeof(uint32_t) * AVK_N_GROUPS * AVK_N_FIELDS);
                if (want_bp_words) { /* the packed form: the repaired region's groups join the spilled ones */
                    const uint32_t ng = fx.bp_off[k + 1] - fx.bp_off[k], at = out->bp_spilled[0];
                    memcpy(out->bp_groups + 4 * (size_t)at, fx.bp.data() + 4 * (size_t)fx.bp_off[k], 16 * (size_t)ng);
                    out->bp_packed[r] = AVK_BP_SPILL | at;
                    out->bp_spilled[0] = at + ng;
                } else if (want_bp && out->bp_off[r + 1] - out->bp_off[r] == fx.bp_off[k + 1] - fx.bp_off[k]) /* the same calls, so the same groups */
                    memcpy(out->bp_groups + 4 * (size_t)out->bp_off[r], fx.bp.data() + 4 * (size_t)fx.bp_off[k], 16 * (size_t)(fx.bp_off[k + 1] - fx.bp_off[k]));
                if (want_seq)
                    for (int q = 0; q < 5; ++q) {
                        uint32_t len = fx.seq_len[5 * k + q];
                        if (len > out->seq_stride[r]) len = out->seq_stride[r];
                        memcpy(out->seq_bytes + out->seq_off[r] + (uint64_t)q * out->seq_stride[r], fx.seq.data() + fx.seq_off[k] + (uint64_t)q * fx.seq_stride[k], len);
                        out->seq_len[5 * r + q] = len;
                    }
            }
            if (out->tally) { /* the sub-batch's sums come from its solved regions only */
                for (int i = 0; i < AVK_N_GROUPS * AVK_N_FIELDS; ++i) out->tally[i] += fx.tally[i];
                out->tally[AVK_TALLY_SOLVED] += fx.tally[AVK_TALLY_SOLVED];
                out->tally[AVK_TALLY_ERRORS] -= fx.tally[AVK_TALLY_SOLVED];
            }
            cap.swap(still);
        }
        memcpy(ctx->last_tiers, keep_tiers, sizeof(keep_tiers)); /* the statistics of the caller's batch, not of the retries */
        ctx->last_lane_solved = keep_lane_solved;
        ctx->last_wide_solved = keep_wide_solved;
        if (big_grown && ctx->big_alloc > big_before && ctx->d_big) { /* give the large slices back */
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(ctx->d_big);
            ctx->d_big = nullptr;
            ctx->big_alloc = 0;
        }
        if (retry_rc) return retry_rc;
    }
    return 0;
}

extern "C" {

int avk_results_download(avk_ctx *ctx, avk_dev_batch *db, avk_result_batch *out) { return results_download_impl(ctx, db, out, nullptr, nullptr); }

/* ---- the asynchronous boundary: one context, batches in flight -------------------------------------------------------------------------------
 * The reference streams its regions through a rayon loop and collects at the end (src/main.rs:251-268); a caller with several batches gets the same here:
 * avk_compare_packed_submit returns when batch k is QUEUED — its arrays crossed (or are crossing) the bus on the context's copy-in stream while batch k - 1 was
 * being solved, its results will cross on the copy-out stream while batch k + 1 is packed — and avk_wait(ticket) returns when the results are in the caller's
 * arrays.  The call blocks once in the middle (the packer's plan comes back to the host), which is where it waits for the batch before it. */
struct avk_ticket {
    int slot = -1;          /* staging slot, -1: the batch was solved synchronously at submit (pageable arrays) */
    int rc = 0;
    avk_dev_batch *db = nullptr;
    avk_result_batch out;
    DownloadLater later;
    int64_t keep_gm = 0, keep_bp = 0;
};

static int stage_slot_prepare(avk_ctx *ctx, avk_ctx::StageSlot &sl, size_t bytes) {
    if (!ctx->copy_in_stream || !ctx->copy_out_stream || !ctx->pack_stream || !ctx->pack_side_stream) {
        /* the four streams of the asynchronous calls, made at the first submit (a context that never submits does not hold their hardware queues: two processes on a
         * GPU with twelve streams each are more than the scheduler maps at a time).  The placeholders of avk_ctx_create are gone by now and their queues would be the
         * first to be handed out again — to these four, which then sit exactly where nothing was meant to (two genomes in flight: 4.8 ms per genome instead of 4.4):
         * the placeholders are made again for the moment, as they were when the order was searched. */
        hipStream_t ph[16];
        int n_ph = 0;
        if (!getenv("AVK_KEEP_SPARES"))
            for (; n_ph < ctx->n_placeholders && n_ph < 16; ++n_ph)
                if (hipStreamCreateWithFlags(&ph[n_ph], hipStreamNonBlocking) != hipSuccess) break;
        hipError_t e = hipSuccess;
        if (!ctx->copy_in_stream) e = hipStreamCreateWithFlags(&ctx->copy_in_stream, hipStreamNonBlocking);
        if (e == hipSuccess && !ctx->copy_out_stream) e = hipStreamCreateWithFlags(&ctx->copy_out_stream, hipStreamNonBlocking);
        if (e == hipSuccess && !ctx->pack_stream) e = hipStreamCreateWithFlags(&ctx->pack_stream, hipStreamNonBlocking);
        if (e == hipSuccess && !ctx->pack_side_stream) e = hipStreamCreateWithFlags(&ctx->pack_side_stream, hipStreamNonBlocking);
        for (int k = 0; k < n_ph; ++k) (void)hipStreamDestroy(ph[k]);
        AVK_HIP(ctx, e);
    }
    if (!ctx->ev_packed) AVK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_packed, hipEventDisableTiming));
    if (!ctx->ev_pool_fence) AVK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pool_fence, hipEventDisableTiming));
    if (!sl.ev_in) {
        AVK_HIP(ctx, hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
        AVK_HIP(ctx, hipEventCreateWithFlags(&sl.ev_unpacked, hipEventDisableTiming));
        AVK_HIP(ctx, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        AVK_HIP(ctx, hipHostMalloc((void **)&sl.h_tally, (size_t)AVK_TALLY_STRIDE * sizeof(uint64_t), hipHostMallocDefault));
    }
    if (sl.bytes < bytes) {
        if (sl.dev) AVK_HIP(ctx, hipFree(sl.dev)); /* (the slot is free: nothing of an earlier batch reads it) */
        sl.dev = nullptr, sl.bytes = 0;
        const size_t want = bytes + bytes / 8 + (1u << 20);
        hipError_t e = hipMalloc((void **)&sl.dev, want);
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? AVK_E_OOM : AVK_E_HIP, "staging slot of %zu bytes: %s", want, hipGetErrorString(e));
        sl.bytes = want;
    }
    return 0;
}

/* the eleven arrays of `batch` into a staging slot: the slot is (re)sized and the copies are queued on the copy-in stream, counts and lengths first as in
 * avk_compare_packed — nothing of this context reads or writes the slot */
static int stage_layout(avk_ctx *ctx, avk_ctx::StageSlot &sl, const avk_packed_batch *batch, PackedOnDevice *pre) {
    const uint64_t n = batch->n_regions, nv = batch->n_variants, alen = batch->allele_bytes_len;
    const bool has_contig = batch->contig_idx != nullptr, has_raw = batch->var_raw_space != nullptr;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t need = 2 * up(n + 16) + 3 * up(nv + 16) + up(n * 4 + 16) + 2 * up(n * 2 + 16) + up(nv * 2 + 16) + up(nv * 4 + 16) + up(alen + 16);
    {
        const int rc = stage_slot_prepare(ctx, sl, need);
        if (rc) return rc;
    }
    memset(pre, 0, sizeof(*pre));
    uint8_t *q = sl.dev;
    auto take = [&](size_t bytes) {
        uint8_t *r = q;
        q += up(bytes + 16);
        return r;
    };
    pre->t_cnt = take(n), pre->q_cnt = take(n), pre->a0_len = take(nv), pre->a1_len = take(nv), pre->start = (uint32_t *)take(n * 4), pre->len = (uint16_t *)take(n * 2);
    pre->contig = (uint16_t *)take(n * 2), pre->rel_pos = (uint16_t *)take(nv * 2), pre->var_type_zyg = take(nv), pre->raw = (uint32_t *)take(nv * 4), pre->alleles = take(alen);
    pre->ready = sl.ev_in;
    const struct { const void *src; void *dst; size_t bytes; } cp[] = {
        {batch->t_cnt, pre->t_cnt, n}, {batch->q_cnt, pre->q_cnt, n}, {batch->a0_len, pre->a0_len, nv}, {batch->a1_len, pre->a1_len, nv}, {batch->start, pre->start, n * 4},
        {batch->len, pre->len, n * 2}, {batch->contig_idx, pre->contig, has_contig ? n * 2 : 0}, {batch->var_rel_pos, pre->rel_pos, nv * 2}, {batch->var_type_zyg, pre->var_type_zyg, nv},
        {batch->var_raw_space, pre->raw, has_raw ? nv * 4 : 0}, {batch->allele_bytes, pre->alleles, alen}};
    hipError_t e = hipSuccess;
    for (const auto &c : cp)
        if (e == hipSuccess && c.bytes && c.src) e = hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, ctx->copy_in_stream);
    if (e == hipSuccess) e = hipEventRecord(sl.ev_in, ctx->copy_in_stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->copy_in_stream);
        return fail(ctx, AVK_E_HIP, "queueing the batch's copies failed: %s", hipGetErrorString(e));
    }
    return 0;
}
static bool packed_inputs_pinned(const avk_packed_batch *batch) {
    const uint64_t n = batch->n_regions, nv = batch->n_variants, alen = batch->allele_bytes_len;
    return is_pinned(batch->t_cnt, n) && is_pinned(batch->q_cnt, n) && is_pinned(batch->a0_len, nv) && is_pinned(batch->a1_len, nv) && is_pinned(batch->start, n * 4) &&
           is_pinned(batch->len, n * 2) && is_pinned(batch->contig_idx, n * 2) && is_pinned(batch->var_rel_pos, nv * 2) && is_pinned(batch->var_type_zyg, nv) &&
           is_pinned(batch->var_raw_space, nv * 4) && is_pinned(batch->allele_bytes, alen);
}

static int submit_impl(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out, avk_ticket **ticket, uint32_t *shared_spill,
                       uint32_t *shared_spill_count);
int avk_compare_packed_submit(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out, avk_ticket **ticket) {
    return submit_impl(ctx, batch, cfg, out, ticket, nullptr, nullptr);
}
/* shared_spill / shared_spill_count: the device list and counter the parts of one split call spill their BASEPAIR groups into (compare_packed_split); with them a
 * batch that returns the packed groups can be queued like any other */
static int submit_impl(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out, avk_ticket **ticket, uint32_t *shared_spill,
                       uint32_t *shared_spill_count) {
    if (!ctx || !batch || !cfg || !out || !ticket || !(out->status || out->region_packed)) return AVK_E_ARG;
    *ticket = nullptr;
    if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = batch->n_regions, nv = batch->n_variants;
    if (n && (!batch->start || !batch->len || !batch->t_cnt || !batch->q_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (nv && (!batch->var_rel_pos || !batch->var_type_zyg || !batch->a0_len || !batch->a1_len || !batch->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    const bool timing = getenv("AVK_TIMING") != nullptr;
    static const auto t_epoch = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_epoch).count(); };
    const double ts0 = now_ms();
    avk_ticket *t = new avk_ticket();
    t->out = *out;
    /* arrays that are not pinned cannot be copied behind the caller's back: such a batch is solved here and now, its ticket is complete */
    bool pinned = packed_inputs_pinned(batch);
    pinned = pinned && is_pinned(out->status, n * 4) && is_pinned(out->region_packed, n * 8) && is_pinned(out->ed_h1, n * 4) && is_pinned(out->ed_h2, n * 4) && is_pinned(out->n_optima, n * 4) &&
             is_pinned(out->type_present, n * 2) && is_pinned(out->var_expected, nv) && is_pinned(out->var_observed, nv) && is_pinned(out->var_class, nv) && is_pinned(out->var_zyg, nv) &&
             is_pinned(out->var_packed, nv) && is_pinned(out->group_metrics, n * AVK_N_GROUPS * AVK_N_FIELDS * 4);
    const bool seq_out = out->seq_bytes && out->seq_len && out->seq_off && out->seq_stride && cfg->enable_sequences;
    int slot = -1;
    const bool bp_words = out->bp_packed && out->bp_spilled && out->bp_groups && !out->bp_off;
    const bool bp_queueable = bp_words && shared_spill && shared_spill_count && is_pinned(out->bp_packed, n * 4);
    const bool can_queue = pinned && !seq_out && (bp_queueable || !((out->bp_off || out->bp_packed) && out->bp_groups));
    for (int i = 0; i < 4 && slot < 0 && can_queue; ++i)
        if (!ctx->stage[i].busy) {
            slot = i;
            break;
        }
    if (slot < 0 && can_queue) {
        delete t;
        return fail(ctx, AVK_E_STATE, "four batches are in flight: avk_wait for one of them first");
    }
    if (slot < 0) { /* solved here and now (it uses no staging slot): a complete ticket on success, no ticket on failure — the caller owns what it is handed */
        const int rc_now = avk_compare_packed(ctx, batch, cfg, out);
        if (rc_now) {
            delete t;
            return rc_now;
        }
        t->rc = 0;
        *ticket = t;
        return AVK_E_OK;
    }
    avk_ctx::StageSlot &sl = ctx->stage[slot];
    PackedOnDevice pre;
    int rc = stage_layout(ctx, sl, batch, &pre);
    if (rc) {
        delete t;
        return rc;
    }
    const double ts1 = now_ms();
    sl.busy = true;
    t->slot = slot;
    t->keep_gm = ctx->emit_group_metrics, t->keep_bp = ctx->emit_bp_groups;
    if (!out->group_metrics) ctx->emit_group_metrics = 0;
    if (bp_queueable) ctx->emit_bp_groups = 1;
    t->later.h_tally = sl.h_tally, t->later.ev_unpacked = sl.ev_unpacked, t->later.ev_done = sl.ev_done;
    t->later.shared_spill = bp_queueable ? shared_spill : nullptr, t->later.shared_spill_count = bp_queueable ? shared_spill_count : nullptr;
    /* Packing on a stream of its own: its kernels stream the batch's arrays through HBM while the solver launches of the batch before are busy with their searches,
     * and the plan's round trip to the host no longer waits for that solve.  Pool buffers stay ordered: what this upload is handed was released either by an upload
     * (on this same stream) or by a batch whose work is over (avk_wait). */
    if (ctx->async_pack_stream) {
        ctx->up_stream = ctx->pack_stream, ctx->up_side = ctx->pack_side_stream;
        if (ctx->pool_fence_pending) { /* a synchronous upload released buffers behind kernels of the context's stream that may still be queued */
            (void)hipStreamWaitEvent(ctx->pack_stream, ctx->ev_pool_fence, 0);
            ctx->pool_fence_pending = false;
        }
    }
    rc = upload_device_packed(ctx, nullptr, nullptr, false, &t->db, nullptr, batch, nullptr, &pre);
    if (!rc && ctx->up_stream) {
        hipError_t e = hipEventRecord(ctx->ev_packed, ctx->up_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_packed, 0);
        if (e != hipSuccess) rc = fail(ctx, AVK_E_HIP, "packed upload: %s", hipGetErrorString(e));
    }
    ctx->up_stream = nullptr, ctx->up_side = nullptr;
    const double ts2 = now_ms();
    if (!rc) rc = avk_compare_resident(ctx, t->db, cfg, nullptr);
    const double ts3 = now_ms();
    if (!rc) rc = results_download_impl(ctx, t->db, out, &t->later, nullptr);
    if (timing)
        fprintf(stderr, "avk submit (slot %d): called at %.3f ms; copies queued +%.3f, packed and planned +%.3f, solver launches queued +%.3f, results queued +%.3f\n", slot, ts0, ts1 - ts0,
                ts2 - ts0, ts3 - ts0, now_ms() - ts0);
    ctx->emit_group_metrics = t->keep_gm, ctx->emit_bp_groups = t->keep_bp;
    if (rc) { /* nothing of this batch stays in flight */
        (void)hipStreamSynchronize(ctx->copy_in_stream);
        (void)hipStreamSynchronize(ctx->pack_stream);
        (void)hipStreamSynchronize(ctx->pack_side_stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->copy_out_stream);
        for (void *p : t->later.temps) pool_release(ctx, p);
        if (t->db) avk_batch_free(ctx, t->db);
        sl.busy = false;
        delete t;
        return rc;
    }
    ctx->last_one_shot = 1;
    *ticket = t;
    return 0;
}

int avk_wait(avk_ctx *ctx, avk_ticket *t) {
    if (!ctx || !t) return AVK_E_ARG;
    if (t->slot < 0) { /* solved at submit */
        const int rc = t->rc;
        delete t;
        return rc;
    }
    (void)hipSetDevice(ctx->device);
    avk_ctx::StageSlot &sl = ctx->stage[t->slot];
    const auto tw0 = std::chrono::steady_clock::now();
    hipError_t e = hipEventSynchronize(sl.ev_done);
    const auto tw1 = std::chrono::steady_clock::now();
    int rc = e == hipSuccess ? 0 : fail(ctx, AVK_E_HIP, "waiting for the batch's results failed: %s", hipGetErrorString(e));
    if (!rc) { /* the tally, the statistics and — should a region have come back AVK_ST_CAPACITY — the retry, as in avk_results_download */
        const int64_t gm = ctx->emit_group_metrics;
        if (!t->out.group_metrics) ctx->emit_group_metrics = 0;
        rc = results_download_impl(ctx, t->db, &t->out, nullptr, sl.h_tally);
        ctx->emit_group_metrics = gm;
    }
    /* the batch's buffers go back to the pool: its work is over (the copies out ran behind its last kernel), so whoever is handed them next may use them at once */
    for (void *p : t->later.temps) pool_release(ctx, p);
    if (t->db) {
        release_pooled(ctx, t->db);
        delete t->db;
    }
    sl.busy = false;
    if (getenv("AVK_TIMING"))
        fprintf(stderr, "avk wait (slot %d): results arrived after %.3f ms, tally and buffers %.3f ms\n", t->slot, std::chrono::duration<double, std::milli>(tw1 - tw0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count());
    delete t;
    return rc;
}

int avk_last_compare_was_one_shot(avk_ctx *ctx) { return ctx ? ctx->last_one_shot : 0; }

} /* extern "C" */
#include "avk_shard_host.inl"
extern "C" {

/* pinned host memory for the caller's batch and result arrays: arrays that live there are copied by DMA, no host pass (pageable arrays go through
 * a pinned bounce buffer that the host threads fill) */
void *avk_host_alloc(avk_ctx *ctx, size_t bytes) {
    if (ctx) (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx) fail(ctx, AVK_E_OOM, "cannot pin %zu bytes of host memory", bytes);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pinned_mutex);
    g_pinned.push_back({(const uint8_t *)p, (const uint8_t *)p + bytes});
    return p;
}

void avk_host_free(avk_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    {
        std::lock_guard<std::mutex> lk(g_pinned_mutex);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].lo == (const uint8_t *)p) {
                g_pinned.erase(g_pinned.begin() + (long)i);
                break;
            }
    }
    (void)hipHostFree(p);
}

/* what a batch of this size will need, ahead of the first call: the pinned bounce buffer for pageable arrays (callers whose arrays come from
 * avk_host_alloc need none) and the device buffers of the pool */
int avk_ctx_reserve(avk_ctx *ctx, uint64_t n_regions, uint64_t n_variants) {
    if (!ctx) return AVK_E_ARG;
    if (n_regions > 0x7FFFFFFFull || n_variants > 0x7FFFFFFFull) return 0;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    return bounce_reserve(ctx, (size_t)n_regions * 52 + (size_t)n_variants * 48 + (1u << 20));
}

/* What a process's FIRST large call pays for beyond the call itself, done ahead of it (a tool: on a thread beside its parsing): the device code is brought in by a
 * launch of nothing, the table of the looked-up class is made for the default branch factor, and the workspaces of a batch of that size are allocated (7 GB for a
 * genome; fresh device memory is scrubbed when it is handed out).  Nothing here changes a result; every step is also made on demand by the call that needs it. */
/* a small call through every kind of launch on a context of its own (the caller's may be taking its reference at this moment): a kernel's FIRST launch in a
 * process costs about a millisecond — the dozen packing kernels of a first whole-genome call took 16 ms instead of 1.5 */
static void warm_kernels(int device) {
    avk_ctx *t = nullptr;
    if (avk_ctx_create(device, &t) || !t) return;
    const char *opts[] = {"lane_min_regions", "lane_min_batch", "emit_group_metrics", "adaptive_ws"}; /* (adaptive_ws 0: with the class C threshold below the packer
                                                                                                          would predict, and this context allocate, tens of GB of slices) */
    for (const char *o : opts) (void)avk_ctx_set_option(t, o, 0);
    (void)avk_ctx_set_option(t, "big_ws_bytes", 8 << 20);
    (void)avk_ctx_set_option(t, "big_waves", 4);
    (void)avk_ctx_set_option(t, "ws_bytes_per_wave", 256 << 10); /* (the warm-up's regions are tiny: 0.4 GB of slices instead of 1.7 — what this context frees at its end is
                                                                   memory the caller's first call may be handed next, and has to wait for while it is scrubbed) */
    (void)avk_ctx_set_option(t, "class_c_nodes_x2", 1000); /* every region the lanes do not take is planned as class C: the wide kernel, the HBM launches */
    (void)avk_ctx_set_option(t, "lane_node_cap", 8);       /* ... and the three-call class hands back (8: the option's smallest value) */
    const uint32_t L = 120, n_contig = 1u << 16;
    std::vector<uint8_t> contig(n_contig);
    uint32_t x = 12345u;
    for (uint32_t i = 0; i < n_contig; ++i) {
        x = x * 1664525u + 1013904223u;
        contig[i] = "ACGT"[x >> 30];
    }
    std::vector<uint32_t> start;
    std::vector<uint16_t> len, rel;
    std::vector<uint8_t> tc, qc, tz, a0l, a1l, alle;
    auto other = [](uint8_t b) { return (uint8_t)(b == 'A' ? 'C' : 'A'); };
    for (uint32_t r = 0; r < 448; ++r) { /* calls per side: 1 (the looked-up class and its neighbours), 2, 3, 5 */
        const uint32_t per = r < 256 ? 1u : (r < 320 ? 2u : (r < 384 ? 3u : 5u)), s0 = 100u + r * 140u;
        start.push_back(s0), len.push_back((uint16_t)L), tc.push_back((uint8_t)per), qc.push_back((uint8_t)per);
        for (uint32_t side = 0; side < 2; ++side)
            for (uint32_t k = 0; k < per; ++k) {
                const uint32_t p = 10u + 20u * k;
                rel.push_back((uint16_t)p);
                tz.push_back((uint8_t)(AVK_VT_SNV | ((r & 1u ? AVK_ZYG_UNPHASED_HET : AVK_ZYG_HOM_ALT) << 4)));
                a0l.push_back(1), a1l.push_back(1);
                alle.push_back(contig[s0 + p]);
                alle.push_back(side && (r & 2u) && k == 0 ? (uint8_t)(other(contig[s0 + p]) == 'C' ? 'G' : 'T') : other(contig[s0 + p]));
            }
    }
    avk_packed_batch b;
    memset(&b, 0, sizeof(b));
    b.n_regions = start.size(), b.start = start.data(), b.len = len.data(), b.t_cnt = tc.data(), b.q_cnt = qc.data();
    b.n_variants = rel.size(), b.var_rel_pos = rel.data(), b.var_type_zyg = tz.data(), b.a0_len = a0l.data(), b.a1_len = a1l.data();
    b.allele_bytes = alle.data(), b.allele_bytes_len = alle.size();
    std::vector<int32_t> status(b.n_regions);
    std::vector<uint32_t> e1(b.n_regions), e2(b.n_regions), no(b.n_regions);
    std::vector<uint16_t> tp(b.n_regions);
    std::vector<uint8_t> ve(b.n_variants), vo(b.n_variants), vc(b.n_variants), vz(b.n_variants);
    std::vector<uint64_t> tally(AVK_TALLY_LEN);
    avk_result_batch out;
    memset(&out, 0, sizeof(out));
    out.status = status.data(), out.ed_h1 = e1.data(), out.ed_h2 = e2.data(), out.n_optima = no.data(), out.type_present = tp.data();
    out.var_expected = ve.data(), out.var_observed = vo.data(), out.var_class = vc.data(), out.var_zyg = vz.data(), out.tally = tally.data();
    const uint8_t *seqs[1] = {contig.data()};
    const uint64_t lens[1] = {n_contig};
    avk_compare_config cfg = {50, 0, 0};
    if (avk_ref_upload(t, 1, seqs, lens) == 0) (void)avk_compare_packed(t, &b, &cfg, &out);
    avk_ctx_destroy(t);
}

int avk_ctx_warmup(avk_ctx *ctx, uint64_t n_regions_hint, uint64_t n_variants_hint) {
    if (!ctx) return AVK_E_ARG;
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = avk_ctx_reserve(ctx, n_regions_hint, n_variants_hint);
    if (rc) return rc;
    warm_kernels(ctx->device);
    unsigned *d_scratch = nullptr;
    struct ScratchGuard { /* released on every way out, the AVK_HIP early returns included */
        unsigned *&p;
        ~ScratchGuard() {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
    } scratch_guard{d_scratch};
    AVK_HIP(ctx, hipMalloc((void **)&d_scratch, (size_t)AVK_TALLY_STRIDE * (AVK_TALLY_COPIES + 1) * sizeof(uint64_t) + AVK_N_COUNTERS * sizeof(uint32_t)));
    AVK_HIP(ctx, hipMemsetAsync(d_scratch, 0, (size_t)AVK_TALLY_STRIDE * (AVK_TALLY_COPIES + 1) * sizeof(uint64_t) + AVK_N_COUNTERS * sizeof(uint32_t), ctx->stream));
    uint64_t *parts = (uint64_t *)d_scratch;
    hipLaunchKernelGGL(avk_tally_reduce, dim3((AVK_TALLY_STRIDE + 63) / 64), dim3(64), 0, ctx->stream, parts, parts + (size_t)AVK_TALLY_STRIDE * AVK_TALLY_COPIES, (uint64_t *)nullptr,
                       (uint32_t *)(parts + (size_t)AVK_TALLY_STRIDE * (AVK_TALLY_COPIES + 1)), (unsigned)AVK_N_COUNTERS, 0u);
    AVK_HIP(ctx, hipGetLastError());
    if (!ctx->lane_attr_set) {
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_lane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        AVK_HIP(ctx, hipFuncSetAttribute((const void *)avk_quad_kernel_wide_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->lane_attr_set = true;
    }
    if (ctx->lane_kernel && ctx->lane_pairs) {
        rc = ensure_pair_table(ctx, 50, ctx->stream);
        if (rc) return rc;
    }
    if (n_regions_hint >= 65536 && ctx->ws_bytes_per_wave > 0) {
        /* The workspaces run_internal sizes for a batch of a genome's size, allocated HERE, on the thread that runs beside the caller's parsing: hipMalloc of the
         * 4.8 + 2.1 GB they were until round 5 took 0.39 s (device memory is scrubbed when it is handed out).  Round 4 took and released as many bytes instead, counting on memory the process has
         * held once to come back at once: it does in two runs of three — in the third the first call's own hipMalloc paid the 0.39 s inside the tool's solve stage
         * (profiles/r05_first_solve.txt: "avk run: workspaces (.. allocated now ..) 387 ms"). */
        const size_t main_blocks = (size_t)ctx->n_cus * 3u < AVK_HBM_BLOCKS_BESIDE_LANES ? (size_t)ctx->n_cus * 3u : (size_t)AVK_HBM_BLOCKS_BESIDE_LANES; /* (a genome has lane launches) */
        const size_t waves = (main_blocks + (size_t)(ctx->hbm_solo_blocks > 0 ? ctx->hbm_solo_blocks : 128) + (size_t)(ctx->hbm_early_blocks > 0 ? ctx->hbm_early_blocks : 64)) * 4u;
        const size_t ws_need = waves * (size_t)ctx->ws_bytes_per_wave;
        if (ws_need > ctx->ws_alloc && (double)ws_need <= (double)ctx->ws_budget_bytes) {
            if (ctx->d_ws) (void)hipFree(ctx->d_ws);
            ctx->d_ws = nullptr, ctx->ws_alloc = 0;
            if (hipMalloc((void **)&ctx->d_ws, ws_need + 256) == hipSuccess) ctx->ws_alloc = ws_need;
            else ctx->d_ws = nullptr, (void)hipGetLastError(); /* (the first call asks again, and reports) */
        }
        const size_t big_need = (size_t)((ctx->big_waves + 3) / 4) * 4u * (size_t)ctx->big_ws_bytes;
        if (ctx->big_ws_bytes > 0 && big_need > ctx->big_alloc) {
            if (ctx->d_big) (void)hipFree(ctx->d_big);
            ctx->d_big = nullptr, ctx->big_alloc = 0;
            if (hipMalloc((void **)&ctx->d_big, big_need + 256) == hipSuccess) ctx->big_alloc = big_need;
            else ctx->d_big = nullptr, (void)hipGetLastError();
        }
    }
    if (n_regions_hint >= 65536 && ctx->device_pack) {
        /* ... and the largest of the batch's own buffers, taken from the context's pool and put back: the first call finds them cached (a cached buffer serves requests
         * of half its size and more, so the hints only have to be about right).  Without them a process's first whole-genome call waited 0.1 s inside one hipMalloc — the
         * 675 MB arena of the region blobs (profiles/r05_first_solve.txt).  AVK_WARM_POOL=1 takes and clears every buffer of the batch instead. */
        if (getenv("AVK_WARM_POOL")) {
            rc = pool_prewarm(ctx, n_regions_hint, n_variants_hint);
            if (rc) return rc;
        } else {
            const size_t n = (size_t)n_regions_hint, nv = (size_t)n_variants_hint;
            const size_t sizes[] = {n * 190, n * 64, n * 64, n * 52, nv * 16, nv * 16, nv * 8, nv * 8, nv * 8}; /* blobs, region records, region info, fast records, call info and slots, positions and allele offsets */
            std::vector<void *> got;
            for (size_t b : sizes) {
                void *p = nullptr;
                if (pool_alloc(ctx, &p, b)) break;
                got.push_back(p);
            }
            for (void *p : got) pool_release(ctx, p);
        }
    }
    AVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int avk_batch_upload_compact(avk_ctx *ctx, const avk_compact_batch *batch, avk_dev_batch **out) {
    if (!ctx || !batch || !out) return AVK_E_ARG;
    *out = nullptr;
    if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    return upload_device_packed(ctx, nullptr, batch, false, out);
}

int avk_batch_upload_packed(avk_ctx *ctx, const avk_packed_batch *batch, avk_dev_batch **out) {
    if (!ctx || !batch || !out) return AVK_E_ARG;
    *out = nullptr;
    if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    return upload_device_packed(ctx, nullptr, nullptr, false, out, nullptr, batch);
}

/* One large call as `parts` batches in flight (context option split_parts; avk_compare_packed below): the regions are independent (src/main.rs:251-268 maps over
 * them), so the batch is cut into ranges of regions — the packed form has no explicit offsets, a range of regions with its calls and allele bytes is a packed batch of
 * its own — that go through the staging slots of the asynchronous boundary: the arrays of part k + 1 cross the bus and are packed while part k is solved, the results of
 * part k cross while part k + 1 is solved.  Same outputs, byte for byte: per-region and per-call arrays land at the parts' places in the caller's arrays, the tallies are
 * added, the parts' spilled BASEPAIR groups form one list (one device buffer, one counter).  Returns -1 when the call does not qualify (the caller then runs it whole). */
static int compare_packed_split(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out) {
    const uint64_t n = batch->n_regions, nv = batch->n_variants;
    int64_t parts = ctx->split_parts;
    if (parts > 4) parts = 4;
    if (parts < 2 || !ctx->async_pack_stream || n / (uint64_t)parts < 4096) return -1;
    if (cfg->enable_sequences && out->seq_bytes) return -1;
    if (out->bp_off || (out->bp_packed && !(out->bp_spilled && out->bp_groups)) || out->group_metrics) return -1;
    for (int i = 0; i < 4; ++i)
        if (ctx->stage[i].busy) return -1; /* batches of the caller's are in flight: the slots are theirs */
    if (!packed_inputs_pinned(batch)) return -1;
    if (!(is_pinned(out->status, n * 4) && is_pinned(out->region_packed, n * 8) && is_pinned(out->ed_h1, n * 4) && is_pinned(out->ed_h2, n * 4) && is_pinned(out->n_optima, n * 4) &&
          is_pinned(out->type_present, n * 2) && is_pinned(out->var_expected, nv) && is_pinned(out->var_observed, nv) && is_pinned(out->var_class, nv) && is_pinned(out->var_zyg, nv) &&
          is_pinned(out->var_packed, nv) && is_pinned(out->bp_packed, n * 4)))
        return -1;
    const bool want_bp = out->bp_packed != nullptr;
    /* where the parts begin: regions in equal shares, their calls and allele bytes by the running sums the packed form implies (host threads: two byte arrays read once) */
    uint64_t r0[5], v0[5], a0[5];
    for (int k = 0; k <= parts; ++k) r0[k] = n * (uint64_t)k / (uint64_t)parts;
    v0[0] = a0[0] = 0;
    {
        const unsigned threads = avk_host_threads();
        for (int k = 0; k < parts; ++k) {
            std::vector<uint64_t> partial(threads + 1, 0);
            avk_parallel_for(r0[k + 1] - r0[k], threads, [&](unsigned t, uint64_t lo, uint64_t hi) {
                uint64_t sum = 0;
                for (uint64_t r = r0[k] + lo; r < r0[k] + hi; ++r) sum += (uint64_t)batch->t_cnt[r] + batch->q_cnt[r];
                partial[t] += sum;
            });
            uint64_t calls = 0;
            for (uint64_t x : partial) calls += x;
            v0[k + 1] = v0[k] + calls;
            if (v0[k + 1] > nv) return fail(ctx, AVK_E_ARG, "packed batch: the call counts sum to more than n_variants %llu", (unsigned long long)nv);
            std::fill(partial.begin(), partial.end(), 0);
            avk_parallel_for(calls, threads, [&](unsigned t, uint64_t lo, uint64_t hi) {
                uint64_t sum = 0;
                for (uint64_t v = v0[k] + lo; v < v0[k] + hi; ++v) sum += (uint64_t)batch->a0_len[v] + batch->a1_len[v];
                partial[t] += sum;
            });
            uint64_t bytes = 0;
            for (uint64_t x : partial) bytes += x;
            a0[k + 1] = a0[k] + bytes;
        }
        if (v0[parts] != nv || a0[parts] != batch->allele_bytes_len)
            return fail(ctx, AVK_E_ARG, "packed batch: the call counts sum to %llu (n_variants %llu), the allele lengths to %llu (allele_bytes_len %llu)", (unsigned long long)v0[parts],
                        (unsigned long long)nv, (unsigned long long)a0[parts], (unsigned long long)batch->allele_bytes_len);
    }
    uint32_t *d_spill = nullptr, *d_count = nullptr;
    if (want_bp) { /* every region has at most 1 + its calls groups */
        int rc = pool_alloc(ctx, (void **)&d_spill, (size_t)(n + nv + 1) * 16);
        if (!rc) rc = pool_alloc(ctx, (void **)&d_count, 256);
        if (!rc && hipMemsetAsync(d_count, 0, 4, ctx->stream) != hipSuccess) rc = fail(ctx, AVK_E_HIP, "split call: %s", hipGetErrorString(hipGetLastError()));
        if (rc) {
            if (d_spill) pool_release(ctx, d_spill);
            if (d_count) pool_release(ctx, d_count);
            return rc;
        }
        out->bp_spilled[0] = 0;
    }
    avk_ticket *tickets[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint64_t> part_tally((size_t)parts * AVK_TALLY_LEN, 0);
    int rc = 0;
    int queued = 0;
    for (int k = 0; k < parts && !rc; ++k) {
        avk_packed_batch pb = *batch;
        pb.n_regions = r0[k + 1] - r0[k], pb.n_variants = v0[k + 1] - v0[k], pb.allele_bytes_len = a0[k + 1] - a0[k];
        pb.contig_idx = batch->contig_idx ? batch->contig_idx + r0[k] : nullptr;
        pb.start = batch->start + r0[k], pb.len = batch->len + r0[k], pb.t_cnt = batch->t_cnt + r0[k], pb.q_cnt = batch->q_cnt + r0[k];
        pb.var_rel_pos = batch->var_rel_pos + v0[k], pb.var_type_zyg = batch->var_type_zyg + v0[k], pb.a0_len = batch->a0_len + v0[k], pb.a1_len = batch->a1_len + v0[k];
        pb.var_raw_space = batch->var_raw_space ? batch->var_raw_space + v0[k] : nullptr;
        pb.allele_bytes = batch->allele_bytes + a0[k];
        avk_result_batch po = *out;
#define AVK_PART(field, at) po.field = out->field ? out->field + (at) : nullptr
        AVK_PART(status, r0[k]), AVK_PART(region_packed, r0[k]), AVK_PART(ed_h1, r0[k]), AVK_PART(ed_h2, r0[k]), AVK_PART(n_optima, r0[k]), AVK_PART(type_present, r0[k]);
        AVK_PART(var_expected, v0[k]), AVK_PART(var_observed, v0[k]), AVK_PART(var_class, v0[k]), AVK_PART(var_zyg, v0[k]), AVK_PART(var_packed, v0[k]);
        AVK_PART(bp_packed, r0[k]);
#undef AVK_PART
        po.tally = part_tally.data() + (size_t)k * AVK_TALLY_LEN;
        rc = submit_impl(ctx, &pb, cfg, &po, &tickets[k], d_spill, d_count);
        if (!rc) queued = k + 1;
    }
    /* everything of every part is on the device's queues; the spilled groups' count and the list itself once the last part's results have crossed */
    if (!rc && queued == parts && want_bp) {
        hipError_t e = tickets[parts - 1]->slot >= 0 ? hipEventSynchronize(ctx->stage[tickets[parts - 1]->slot].ev_done) : hipSuccess;
        for (int k = 0; k + 1 < parts && e == hipSuccess; ++k)
            if (tickets[k]->slot >= 0) e = hipEventSynchronize(ctx->stage[tickets[k]->slot].ev_done);
        uint32_t spilled = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&spilled, d_count, 4, hipMemcpyDeviceToHost, ctx->copy_out_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_out_stream);
        if (e == hipSuccess && spilled) e = hipMemcpyAsync(out->bp_groups, d_spill, (size_t)spilled * 16, hipMemcpyDeviceToHost, ctx->copy_out_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_out_stream);
        if (e != hipSuccess) rc = fail(ctx, AVK_E_HIP, "split call: %s", hipGetErrorString(e));
        out->bp_spilled[0] = spilled; /* (a part's capacity repair, should there be one, appends behind these: avk_wait below) */
    }
    uint64_t tiers[5] = {0, 0, 0, 0, 0}, lane_solved = 0, wide_solved = 0;
    for (int k = 0; k < queued; ++k) {
        const int rw = avk_wait(ctx, tickets[k]);
        if (rw && !rc) rc = rw;
        for (int i = 0; i < 5; ++i) tiers[i] += ctx->last_tiers[i];
        lane_solved += ctx->last_lane_solved, wide_solved += ctx->last_wide_solved;
    }
    if (d_spill) pool_release(ctx, d_spill);
    if (d_count) pool_release(ctx, d_count);
    if (rc) return rc;
    memcpy(ctx->last_tiers, tiers, sizeof(tiers));
    ctx->last_lane_solved = lane_solved, ctx->last_wide_solved = wide_solved;
    if (out->tally) {
        memset(out->tally, 0, AVK_TALLY_LEN * sizeof(uint64_t));
        for (int k = 0; k < parts; ++k)
            for (int i = 0; i < AVK_TALLY_LEN; ++i) out->tally[i] += part_tally[(size_t)k * AVK_TALLY_LEN + i];
    }
    ctx->last_one_shot = 1;
    return 0;
}

int avk_compare_packed(avk_ctx *ctx, const avk_packed_batch *batch, const avk_compare_config *cfg, avk_result_batch *out) {
    if (!ctx || !batch || !cfg || !out || !(out->status || out->region_packed)) return AVK_E_ARG;
    ctx->last_one_shot = 0;
    if (ctx->split_parts > 1 && ctx->d_ref) {
        const int rs = compare_packed_split(ctx, batch, cfg, out);
        if (rs >= 0) return rs;
    }
    avk_dev_batch *db = nullptr;
    const int64_t keep_gm = ctx->emit_group_metrics, keep_bp = ctx->emit_bp_groups;
    if (!out->group_metrics) ctx->emit_group_metrics = 0;
    if ((out->bp_off || (out->bp_packed && out->bp_spilled)) && out->bp_groups) ctx->emit_bp_groups = 1;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = avk_batch_upload_packed(ctx, batch, &db);
    const auto t1 = std::chrono::steady_clock::now();
    if (!rc) rc = avk_compare_resident(ctx, db, cfg, nullptr);
    const auto t2 = std::chrono::steady_clock::now();
    if (!rc) rc = avk_results_download(ctx, db, out);
    const auto t3 = std::chrono::steady_clock::now();
    ctx->emit_group_metrics = keep_gm, ctx->emit_bp_groups = keep_bp;
    if (db) {
        ctx->last_one_shot = 1;
        avk_batch_free(ctx, db);
    }
    if (getenv("AVK_TIMING")) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "avk compare packed: upload %.3f ms, launches %.3f ms, download %.3f ms, free %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3),
                ms(t3, std::chrono::steady_clock::now()));
        if (!rc && ctx->ev_tl[4] && ctx->ev_valid) { /* the same call on the device's clock, free-running (no profiler): events on the context's stream */
            float c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
            if (hipEventElapsedTime(&c1, ctx->ev_tl[0], ctx->ev_tl[1]) == hipSuccess && hipEventElapsedTime(&c2, ctx->ev_tl[0], ctx->ev_tl[2]) == hipSuccess &&
                hipEventElapsedTime(&c3, ctx->ev_tl[0], ctx->ev_tl[3]) == hipSuccess && hipEventElapsedTime(&c4, ctx->ev_tl[0], ctx->ev0) == hipSuccess &&
                hipEventElapsedTime(&c5, ctx->ev_tl[0], ctx->ev1) == hipSuccess && hipEventElapsedTime(&c6, ctx->ev_tl[0], ctx->ev_tl[4]) == hipSuccess)
                fprintf(stderr, "avk compare packed, device clock from the first copy: copies in done %.3f ms, work order done %.3f, record writers done %.3f, solver launches %.3f .. %.3f, results out %.3f\n",
                        c1, c2, c3, c4, c5, c6);
            else
                (void)hipGetLastError();
            /* where each chain of the launch graph ended, from the first solver launch (events of this call only when the batch used the chain) */
            const struct { const char *what; hipEvent_t ev; } chains[] = {{"LDS solo", ctx->ev_join}, {"HBM solo", ctx->ev_join2}, {"wide", ctx->ev_wide}, {"lane stream 1", ctx->ev_lane_join},
                {"lane stream 2", ctx->ev_lane_join2}, {"lane stream 3", ctx->ev_lane_join3}, {"lane stream 4", ctx->ev_lane_join4}, {"early hand-backs", ctx->ev_lane_early},
                {"hand-back launches", ctx->ev_lane_done}};
            fprintf(stderr, "avk compare packed, chains end (ms after the first solver launch):");
            for (const auto &c : chains) {
                float t = 0;
                if (c.ev && hipEventElapsedTime(&t, ctx->ev0, c.ev) == hipSuccess) fprintf(stderr, " %s %.3f;", c.what, t);
                else (void)hipGetLastError();
            }
            fprintf(stderr, " all %.3f\n", c5 - c4);
        }
    }
    return rc;
}

int avk_compare_compact(avk_ctx *ctx, const avk_compact_batch *batch, const avk_compare_config *cfg, avk_result_batch *out) {
    if (!ctx || !batch || !cfg || !out || !(out->status || out->region_packed)) return AVK_E_ARG;
    ctx->last_one_shot = 0;
    avk_dev_batch *db = nullptr;
    const int64_t keep_gm = ctx->emit_group_metrics, keep_bp = ctx->emit_bp_groups;
    if (!out->group_metrics) ctx->emit_group_metrics = 0;
    if ((out->bp_off || (out->bp_packed && out->bp_spilled)) && out->bp_groups) ctx->emit_bp_groups = 1;
    int rc = avk_batch_upload_compact(ctx, batch, &db);
    if (!rc) rc = avk_compare_resident(ctx, db, cfg, nullptr);
    if (!rc) rc = avk_results_download(ctx, db, out);
    ctx->emit_group_metrics = keep_gm, ctx->emit_bp_groups = keep_bp;
    if (db) {
        ctx->last_one_shot = 1;
        avk_batch_free(ctx, db);
    }
    return rc;
}

int avk_compare_batch(avk_ctx *ctx, const avk_region_batch *batch, const avk_compare_config *cfg, avk_result_batch *out) {
    if (!ctx || !batch || !cfg || !out || !(out->status || out->region_packed)) return AVK_E_ARG;
    if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    ctx->last_one_shot = 0;
    avk_dev_batch *db = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    /* per-region metric blocks only when the caller has an array for them */
    const int64_t keep_gm = ctx->emit_group_metrics, keep_bp = ctx->emit_bp_groups;
    if (!out->group_metrics) ctx->emit_group_metrics = 0;
    if ((out->bp_off || (out->bp_packed && out->bp_spilled)) && out->bp_groups) ctx->emit_bp_groups = 1;
    int rc = avk_batch_upload(ctx, batch, &db);
    const auto t1 = std::chrono::steady_clock::now();
    if (!rc) rc = avk_compare_resident(ctx, db, cfg, nullptr);
    const auto t2 = std::chrono::steady_clock::now();
    if (!rc) rc = avk_results_download(ctx, db, out);
    const auto t3 = std::chrono::steady_clock::now();
    ctx->emit_group_metrics = keep_gm, ctx->emit_bp_groups = keep_bp;
    if (db) {
        ctx->last_one_shot = db->dev_packed ? 1 : 0;
        avk_batch_free(ctx, db);
    }
    if (getenv("AVK_TIMING")) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "avk compare batch: upload %.3f ms, launches %.3f ms, download %.3f ms, free %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3),
                ms(t3, std::chrono::steady_clock::now()));
    }
    return rc;
}

int avk_dwfa_script_batch(avk_ctx *ctx, int engine, uint32_t n_scripts, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *base_off, const uint64_t *other_off,
                          const uint64_t *step_off, const uint8_t *step_op, const uint32_t *step_blen, const uint32_t *step_olen, uint32_t *step_ed,
                          int32_t *step_status, uint32_t wf_cap, uint32_t *final_wf, uint32_t *final_wf_len) {
    if (!ctx || !bytes || !base_off || !other_off || !step_off || !step_op || !step_blen || !step_olen || !step_ed || !step_status) return AVK_E_ARG;
    if (engine != 0 && engine != 1) return fail(ctx, AVK_E_ARG, "engine must be 0 (wave per script) or 1 (lane per script)");
    if (n_scripts == 0) return 0;
    if (wf_cap < 3) return fail(ctx, AVK_E_ARG, "wf_cap must be at least 3");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_steps = step_off[n_scripts];
    for (uint32_t s = 0; s < n_scripts; ++s) {
        if (step_off[s + 1] < step_off[s] || base_off[s] > n_bytes || other_off[s] > n_bytes) return fail(ctx, AVK_E_ARG, "script %u: offsets out of range", s);
        for (uint64_t k = step_off[s]; k < step_off[s + 1]; ++k)
            if (step_op[k] > 1 || base_off[s] + step_blen[k] > n_bytes || other_off[s] + step_olen[k] > n_bytes)
                return fail(ctx, AVK_E_ARG, "script %u: a step reads past the byte arena", s);
    }
    AvkDwfaArgs a;
    memset(&a, 0, sizeof(a));
    a.wf_cap = wf_cap;
    a.n_scripts = n_scripts;
    std::vector<void *> allocs;
    auto up = [&](const void *src, size_t nbytes, void **dst) -> hipError_t {
        hipError_t e = hipMalloc(dst, nbytes ? nbytes : 16);
        if (e != hipSuccess) return e;
        allocs.push_back(*dst);
        return nbytes && src ? hipMemcpyAsync(*dst, src, nbytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
    };
    hipError_t e = up(bytes, n_bytes, (void **)&a.bytes);
    if (e == hipSuccess) e = up(base_off, n_scripts * sizeof(uint64_t), (void **)&a.base_off);
    if (e == hipSuccess) e = up(other_off, n_scripts * sizeof(uint64_t), (void **)&a.other_off);
    if (e == hipSuccess) e = up(step_off, (n_scripts + 1) * sizeof(uint64_t), (void **)&a.step_off);
    if (e == hipSuccess) e = up(step_op, n_steps, (void **)&a.step_op);
    if (e == hipSuccess) e = up(step_blen, n_steps * sizeof(uint32_t), (void **)&a.step_blen);
    if (e == hipSuccess) e = up(step_olen, n_steps * sizeof(uint32_t), (void **)&a.step_olen);
    if (e == hipSuccess) e = up(nullptr, n_steps * sizeof(uint32_t), (void **)&a.step_ed);
    if (e == hipSuccess) e = up(nullptr, n_steps * sizeof(int32_t), (void **)&a.step_status);
    if (e == hipSuccess && final_wf) e = up(nullptr, (size_t)n_scripts * wf_cap * sizeof(uint32_t), (void **)&a.final_wf);
    if (e == hipSuccess && final_wf_len) e = up(nullptr, (size_t)n_scripts * sizeof(uint32_t), (void **)&a.final_wf_len);
    if (e == hipSuccess && engine == 0) e = up(nullptr, (size_t)n_scripts * wf_cap * sizeof(uint32_t), (void **)&a.ws);
    if (e == hipSuccess) {
        if (engine == 0) hipLaunchKernelGGL(avk_dwfa_wave_kernel, dim3((n_scripts + 3) / 4), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(avk_dwfa_lane_kernel, dim3((n_scripts + 63) / 64), dim3(64), (size_t)(3 * (AVK_DWFA_LANE_W + 1) + 3 * ((2 * AVK_DWFA_LANE_ED + 2 + 3) / 4)) * 256, ctx->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(step_ed, a.step_ed, n_steps * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(step_status, a.step_status, n_steps * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && final_wf) e = hipMemcpyAsync(final_wf, a.final_wf, (size_t)n_scripts * wf_cap * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && final_wf_len) e = hipMemcpyAsync(final_wf_len, a.final_wf_len, (size_t)n_scripts * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    for (void *p : allocs) (void)hipFree(p);
    if (e != hipSuccess) return fail(ctx, AVK_E_HIP, "avk_dwfa_script_batch failed: %s", hipGetErrorString(e));
    return 0;
}

int avk_label_tallies(avk_ctx *ctx, avk_dev_batch *db, uint32_t n_labels, const uint64_t *label_off, const uint32_t *label_idx, uint64_t *out) {
    if (!ctx || !db || !label_off || !out || (label_off[db->n_regions] && !label_idx)) return AVK_E_ARG;
    if (n_labels == 0) return 0;
    if (!db->d_gm) return fail(ctx, AVK_E_STATE, "the batch has no per-region metric blocks on the device: set emit_group_metrics before avk_compare_resident");
    AVK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = db->n_regions, n_idx = label_off[n];
    for (uint64_t r = 0; r < n; ++r)
        if (label_off[r + 1] < label_off[r]) return fail(ctx, AVK_E_ARG, "label_off must not decrease");
    for (uint64_t q = 0; q < n_idx; ++q)
        if (label_idx[q] >= n_labels) return fail(ctx, AVK_E_ARG, "label index %u of %u", label_idx[q], n_labels);
    unsigned long long *d_off = nullptr, *d_out = nullptr;
    uint32_t *d_idx = nullptr;
    int rc = dev_alloc(ctx, &d_off, (size_t)n + 1);
    if (!rc) rc = dev_alloc(ctx, &d_idx, (size_t)n_idx);
    if (!rc) rc = dev_alloc(ctx, &d_out, (size_t)n_labels * AVK_TALLY_LEN);
    std::vector<uint64_t> host((size_t)n_labels * AVK_TALLY_LEN, 0);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(d_off, label_off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_idx) e = hipMemcpyAsync(d_idx, label_idx, n_idx * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, host.size() * sizeof(uint64_t), ctx->stream);
        uint32_t blocks = (uint32_t)ctx->n_cus * 4u;
        if ((uint64_t)blocks * 4 > n) blocks = (uint32_t)((n + 3) / 4);
        const bool timing = getenv("AVK_TIMING") != nullptr;
        if (timing && e == hipSuccess) e = hipEventRecord(ctx->ev0, ctx->stream);
        for (uint32_t lo = 0; lo < n_labels && e == hipSuccess && n; lo += AVK_LABEL_BLOCK) {
            const uint32_t hi = lo + AVK_LABEL_BLOCK < n_labels ? lo + AVK_LABEL_BLOCK : n_labels;
            hipLaunchKernelGGL(avk_label_tally_kernel, dim3(blocks), dim3(256), 0, ctx->stream, db->d_gm, db->d_region_out, d_off, d_idx, (uint32_t)n, lo, hi, d_out);
            e = hipGetLastError();
        }
        if (timing && e == hipSuccess) {
            e = hipEventRecord(ctx->ev1, ctx->stream);
            if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
            fprintf(stderr, "avk label tallies: %u labels, %llu regions, %llu list entries: kernels %.3f ms (%.1f GB/s of per-region blocks per launch)\n", n_labels,
                    (unsigned long long)n, (unsigned long long)n_idx, ms,
                    ms > 0 ? (double)n * AVK_GM_WORDS * 4 * ((n_labels + AVK_LABEL_BLOCK - 1) / AVK_LABEL_BLOCK) / (ms * 1e6) : 0.0);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d_out, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (d_off) (void)hipFree(d_off);
    if (d_idx) (void)hipFree(d_idx);
    if (d_out) (void)hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, AVK_E_HIP, "label tallies failed: %s", hipGetErrorString(e));
    for (size_t k = 0; k < host.size(); ++k) out[k] += host[k];
    return 0;
}

/* solve_merge_region's pairwise test (merge_solver.rs:128-147) for every region of the batch: the "truth"
 * range is input i, the "query" range input j */
int avk_optimize_pairs_batch(avk_ctx *ctx, const avk_region_batch *batch, uint32_t max_branch_factor, int32_t *status, uint8_t *is_exact_match) {
    if (!ctx || !batch || !status || !is_exact_match) return AVK_E_ARG;
    if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
    avk_compare_config cfg;
    cfg.max_branch_factor = max_branch_factor;
    cfg.enable_sequences = 0;
    cfg.enable_exact_shortcut = 0;
    avk_dev_batch *db = nullptr;
    int rc = upload_internal(ctx, batch, true, &db);
    if (rc) return rc;
    const int64_t keep = ctx->emit_group_metrics;
    ctx->emit_group_metrics = 0;
    rc = run_internal(ctx, db, &cfg, nullptr, 1);
    ctx->emit_group_metrics = keep;
    if (!rc && db->dev_packed) { /* status and the exact-match flag in the caller's layout come from dp_unpack */
        avk_result_batch ro;
        memset(&ro, 0, sizeof(ro));
        ro.status = status;
        std::vector<uint64_t> tally((size_t)AVK_TALLY_STRIDE);
        rc = download_device_packed(ctx, db, &ro, is_exact_match, tally.data());
        ctx->last_one_shot = rc == 0;
        memcpy(ctx->last_tiers, tally.data() + AVK_TALLY_LEN, 5 * sizeof(uint64_t));
        ctx->last_lane_solved = tally[AVK_TALLY_LANE_SOLVED];
        ctx->last_wide_solved = tally[AVK_TALLY_WIDE_SOLVED];
    } else if (!rc) {
        std::vector<uint32_t> rout(db->n_regions * 4 + 4);
        hipError_t e = hipMemcpyAsync(rout.data(), db->d_region_out, db->n_regions * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, AVK_E_HIP, "pairs download failed: %s", hipGetErrorString(e));
        else
            for (uint64_t r = 0; r < db->n_regions; ++r) {
                status[r] = (int32_t)rout[4 * r];
                is_exact_match[r] = status[r] == 0 && rout[4 * r + 1] ? 1 : 0;
            }
    }
    avk_batch_free(ctx, db);
    return rc;
}

/* solve_merge_region's classification (merge_solver.rs:149-199) on top of the pair matrix */
int avk_merge_classify(uint64_t n_regions, uint32_t k, const uint32_t *in_cnt, const uint8_t *has_unknown_zyg, const int32_t *pair_status,
                       const uint8_t *pair_exact, const avk_merge_config *cfg, int32_t *status, uint8_t *classification, uint64_t *members) {
    if (!in_cnt || !cfg || !status || !classification || !members || (n_regions && (!pair_status || !pair_exact))) return AVK_E_ARG;
    if (k < 1 || k > 64) return AVK_E_ARG;
    const uint64_t ppr = (uint64_t)k * (k - 1) / 2;
    for (uint64_t m = 0; m < n_regions; ++m) {
        const int32_t *pst = pair_status + m * ppr;
        const uint8_t *pex = pair_exact + m * ppr;
        avk::dp::merge_classify_one<64>(k, in_cnt + m * k, has_unknown_zyg && has_unknown_zyg[m], [pst, pex](uint64_t p, int32_t &st, bool &ex) {
            st = pst[p];
            ex = pex[p] != 0;
        }, cfg->no_conflict_enabled, cfg->majority_voting_enabled, cfg->conflict_selection, status + m, classification + m, members + m);
    }
    return 0;
}

static bool merge_on_device(const avk_ctx *ctx, uint64_t n_regions, uint32_t k) {
    return ctx->device_pack && k >= 2 && k <= (uint32_t)avk::dp::DP_MERGE_KMAX && n_regions && n_regions * (uint64_t)(k * (k - 1) / 2) <= 0x7FFFFFFFull;
}
static int merge_batch_internal(avk_ctx *ctx, const avk_multi_batch *mb, const avk_packed_multi_batch *pm, const avk_merge_config *cfg, int32_t *status, uint8_t *classification,
                                uint64_t *members);

int avk_merge_packed(avk_ctx *ctx, const avk_packed_multi_batch *pm, const avk_merge_config *cfg, int32_t *status, uint8_t *classification, uint64_t *members) {
    if (!ctx || !pm || !cfg || !status || !classification || !members) return AVK_E_ARG;
    const uint32_t k = pm->n_inputs;
    if (k < 1 || k > 64) return fail(ctx, AVK_E_ARG, "n_inputs must be in [1, 64]");
    const uint64_t n = pm->n_regions, nv = pm->n_variants;
    if (n && (!pm->start || !pm->len || !pm->in_cnt)) return fail(ctx, AVK_E_ARG, "region arrays missing");
    if (nv && (!pm->var_rel_pos || !pm->var_type_zyg || !pm->a0_len || !pm->a1_len || !pm->allele_bytes)) return fail(ctx, AVK_E_ARG, "variant arrays missing");
    avk_multi_batch mb;
    memset(&mb, 0, sizeof(mb));
    mb.n_regions = n, mb.n_inputs = k, mb.n_variants = nv, mb.allele_bytes = pm->allele_bytes, mb.allele_bytes_len = pm->allele_bytes_len;
    if (merge_on_device(ctx, n, k)) return merge_batch_internal(ctx, &mb, pm, cfg, status, classification, members);
    /* no device path for this batch (more than DP_MERGE_KMAX inputs, an empty batch, device_pack = 0): the wide form, made here, through avk_merge_batch */
    std::vector<uint64_t> start(n), end(n), in_off(n * k), pos(nv), a0_off(nv), a1_off(nv);
    std::vector<uint32_t> contig(n), in_cnt(n * k), a0_len(nv), a1_len(nv), raw(nv);
    std::vector<uint8_t> type(nv + 1), zyg(nv + 1);
    uint64_t v = 0, ab = 0;
    for (uint64_t m = 0; m < n; ++m) {
        contig[m] = pm->contig_idx ? pm->contig_idx[m] : 0u, start[m] = pm->start[m], end[m] = (uint64_t)pm->start[m] + pm->len[m];
        for (uint32_t i = 0; i < k; ++i) {
            in_off[m * k + i] = v, in_cnt[m * k + i] = pm->in_cnt[m * k + i];
            for (uint32_t q = 0; q < pm->in_cnt[m * k + i]; ++q, ++v) {
                if (v >= nv) return fail(ctx, AVK_E_ARG, "packed batch: the call counts sum to more than n_variants %llu", (unsigned long long)nv);
                pos[v] = start[m] + pm->var_rel_pos[v];
            }
        }
    }
    if (v != nv) return fail(ctx, AVK_E_ARG, "packed batch: the call counts sum to %llu (n_variants %llu)", (unsigned long long)v, (unsigned long long)nv);
    for (v = 0; v < nv; ++v) {
        a0_off[v] = ab, a0_len[v] = pm->a0_len[v], a1_off[v] = ab + a0_len[v], a1_len[v] = pm->a1_len[v], ab += (uint64_t)a0_len[v] + a1_len[v];
        raw[v] = pm->var_raw_space ? pm->var_raw_space[v] : (a0_len[v] > a1_len[v] ? a0_len[v] : a1_len[v]);
        type[v] = pm->var_type_zyg[v] & 15u, zyg[v] = pm->var_type_zyg[v] >> 4;
    }
    if (ab != pm->allele_bytes_len && nv) return fail(ctx, AVK_E_ARG, "packed batch: the allele lengths sum to %llu (allele_bytes_len %llu)", (unsigned long long)ab, (unsigned long long)pm->allele_bytes_len);
    mb.contig_idx = contig.data(), mb.start = start.data(), mb.end = end.data(), mb.in_off = in_off.data(), mb.in_cnt = in_cnt.data(), mb.var_pos = pos.data(), mb.var_type = type.data(),
    mb.var_zyg = zyg.data(), mb.var_raw_space = raw.data(), mb.a0_off = a0_off.data(), mb.a0_len = a0_len.data(), mb.a1_off = a1_off.data(), mb.a1_len = a1_len.data();
    return avk_merge_batch(ctx, &mb, cfg, status, classification, members);
}

int avk_merge_batch(avk_ctx *ctx, const avk_multi_batch *mb, const avk_merge_config *cfg, int32_t *status, uint8_t *classification, uint64_t *members) {
    return merge_batch_internal(ctx, mb, nullptr, cfg, status, classification, members);
}

static int merge_batch_internal(avk_ctx *ctx, const avk_multi_batch *mb, const avk_packed_multi_batch *pm, const avk_merge_config *cfg, int32_t *status, uint8_t *classification,
                                uint64_t *members) {
    if (!ctx || !mb || !cfg || !status || !classification || !members) return AVK_E_ARG;
    const uint32_t k = mb->n_inputs;
    if (k < 1 || k > 64) return fail(ctx, AVK_E_ARG, "n_inputs must be in [1, 64]");
    if (merge_on_device(ctx, mb->n_regions, k)) {
        /* all of solve_merge_region on the device: the MultiRegions as they are over PCIe, one region per input pair made by a kernel, the pair solve (mode 1),
         * the decision on top of the pair matrix by a kernel; status / classification / members come back */
        if (!ctx->d_ref) return fail(ctx, AVK_E_STATE, "avk_ref_upload has not been called");
        AVK_HIP(ctx, hipSetDevice(ctx->device));
        avk_dev_batch *db = nullptr;
        int rc = upload_device_packed(ctx, nullptr, nullptr, true, &db, mb, nullptr, pm);
        if (rc) return rc;
        avk_compare_config pcfg;
        pcfg.max_branch_factor = cfg->max_branch_factor, pcfg.enable_sequences = 0, pcfg.enable_exact_shortcut = 0;
        const int64_t keep = ctx->emit_group_metrics;
        ctx->emit_group_metrics = 0;
        rc = run_internal(ctx, db, &pcfg, nullptr, 1);
        ctx->emit_group_metrics = keep;
        const uint64_t nm = mb->n_regions;
        void *d_st = nullptr, *d_cl = nullptr, *d_mem = nullptr;
        if (!rc) rc = pool_alloc(ctx, &d_st, (nm + 1) * 4);
        if (!rc) rc = pool_alloc(ctx, &d_cl, nm + 16);
        if (!rc) rc = pool_alloc(ctx, &d_mem, (nm + 1) * 8);
        if (!rc) {
            avk::dp::DpMerge c;
            memset(&c, 0, sizeof(c));
            c.region_out = db->d_region_out, c.in_off = db->d_m_in_off, c.in_cnt = db->d_m_in_cnt, c.var_zyg = db->d_in_zyg, c.n_multi = nm, c.n_variants = mb->n_variants, c.k = k,
            c.ppr = k * (k - 1) / 2, c.no_conflict_enabled = cfg->no_conflict_enabled, c.majority_voting_enabled = cfg->majority_voting_enabled, c.conflict_selection = cfg->conflict_selection;
            c.status = (int32_t *)d_st, c.classification = (uint8_t *)d_cl, c.members = (uint64_t *)d_mem;
            hipLaunchKernelGGL(avk_dp_merge_classify_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, ctx->stream, c);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = fail(ctx, AVK_E_HIP, "merge classification failed: %s", hipGetErrorString(e));
        }
        if (!rc) {
            std::vector<CopySeg> segs = {{status, d_st, nm * 4}, {classification, d_cl, nm}, {members, d_mem, nm * 8}};
            CopyOut co;
            rc = copy_out(ctx, segs, &co);
            if (!rc) rc = finish_copy_out(ctx, co);
        }
        (void)hipStreamSynchronize(ctx->stream);
        pool_release(ctx, d_st), pool_release(ctx, d_cl), pool_release(ctx, d_mem);
        ctx->last_one_shot = rc == 0;
        avk_batch_free(ctx, db);
        return rc;
    }
    const uint64_t n = mb->n_regions, ppr = (uint64_t)k * (k - 1) / 2, np = n * ppr;
    /* one CompareRegion-shaped item per (i < j) pair: input i plays the truth side, input j the query side */
    std::vector<uint64_t> rid(np), st(np), en(np), t_off(np), q_off(np);
    std::vector<uint32_t> cidx(np), t_cnt(np), q_cnt(np);
    std::vector<uint8_t> unknown(n, 0);
    uint64_t p = 0;
    for (uint64_t m = 0; m < n; ++m) {
        for (uint32_t i = 0; i < k; ++i) {
            const uint64_t off = mb->in_off[m * k + i];
            const uint32_t cnt = mb->in_cnt[m * k + i];
            if (off > mb->n_variants || cnt > mb->n_variants - off) return fail(ctx, AVK_E_ARG, "variant range of a region exceeds n_variants");
            for (uint32_t v = 0; v < cnt; ++v)
                if (mb->var_zyg[off + v] == AVK_ZYG_UNKNOWN) unknown[m] = 1;
        }
        for (uint32_t i = 0; i < k; ++i)
            for (uint32_t j = i + 1; j < k; ++j, ++p) {
                rid[p] = mb->region_id ? mb->region_id[m] : m;
                cidx[p] = mb->contig_idx ? mb->contig_idx[m] : 0;
                st[p] = mb->start[m];
                en[p] = mb->end[m];
                t_off[p] = mb->in_off[m * k + i];
                t_cnt[p] = mb->in_cnt[m * k + i];
                q_off[p] = mb->in_off[m * k + j];
                q_cnt[p] = mb->in_cnt[m * k + j];
            }
    }
    avk_region_batch b;
    memset(&b, 0, sizeof(b));
    b.n_regions = np;
    b.region_id = rid.data();
    b.contig_idx = cidx.data();
    b.start = st.data();
    b.end = en.data();
    b.t_off = t_off.data();
    b.t_cnt = t_cnt.data();
    b.q_off = q_off.data();
    b.q_cnt = q_cnt.data();
    b.n_variants = mb->n_variants;
    b.var_pos = mb->var_pos;
    b.var_type = mb->var_type;
    b.var_zyg = mb->var_zyg;
    b.var_raw_space = mb->var_raw_space;
    b.a0_off = mb->a0_off;
    b.a0_len = mb->a0_len;
    b.a1_off = mb->a1_off;
    b.a1_len = mb->a1_len;
    b.allele_bytes = mb->allele_bytes;
    b.allele_bytes_len = mb->allele_bytes_len;
    std::vector<int32_t> pst(np ? np : 1, 0);
    std::vector<uint8_t> pex(np ? np : 1, 0);
    if (np) {
        const int rc = avk_optimize_pairs_batch(ctx, &b, cfg->max_branch_factor, pst.data(), pex.data());
        if (rc) return rc;
    }
    return avk_merge_classify(n, k, mb->in_cnt, unknown.data(), pst.data(), pex.data(), cfg, status, classification, members);
}

} /* extern "C" */
