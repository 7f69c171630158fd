/*
 * avk_dwfa_script.inl — the dynamic wavefront aligner of the kernels, driven directly: a batch of SCRIPTS, each a pair of byte strings
 * and a list of DWFALite calls (update / finalize on prefixes of the two strings, reference src/dwfa/dynamic_wfa.rs:68-84, :183-198),
 * one script per wavefront (engine 0: the lane-group aligner of avk_solver.inl) or one script per lane (engine 1: the 2-bit aligner of
 * avk_lane.inl).  This is how the reference's own DWFA known-answer tests (dynamic_wfa.rs:283-468, the 5,278-edit vector included)
 * reach the device code: tests/test_dwfa_scripts.py runs them through the emulator and, -m gpu, through avk_dwfa_script_batch.
 */
#ifndef AVK_DWFA_SCRIPT_INL
#define AVK_DWFA_SCRIPT_INL

#include "avk_lane.inl"
#include "avk_solver.inl"

struct AvkDwfaArgs {
    const uint8_t *bytes;
    const uint64_t *base_off, *other_off; /* [n] start of the baseline / other string of a script */
    const uint64_t *step_off;             /* [n + 1] the script's steps */
    const uint8_t *step_op;               /* 0 update, 1 finalize */
    const uint32_t *step_blen, *step_olen; /* prefix lengths handed to the call */
    uint32_t *step_ed;                    /* out: edit_distance() after the call */
    int32_t *step_status;                 /* out: 0, 2 = AlreadyFinalized (:69-71, :184-186), AVK_ST_CAPACITY, AVK_ST_INVALID_INPUT (engine 1: not ACGT / too long) */
    uint32_t *final_wf;                   /* out, optional: [n][wf_cap] the wavefront after the last call */
    uint32_t *final_wf_len;               /* out, optional: [n] */
    uint32_t *ws;                         /* engine 0: [n][wf_cap] wavefront workspace */
    uint32_t wf_cap, n_scripts;
};

#define AVK_DWFA_LANE_W 12u   /* engine 1: strings of at most 192 bases */
#define AVK_DWFA_LANE_ED 40u  /* engine 1: wavefront cap */

namespace avk {

/* engine 0: the whole wave works on script s */
AVK_DEV void dwfa_script_wave(const AvkDwfaArgs &a, u32 s) {
    u32 *wf = a.ws + (u64)s * a.wf_cap;
    u32 ed = 0;
    bool finalized = false;
    st32(wf, 0);
    wv_sync();
    const u8 *B = a.bytes + a.base_off[s], *O = a.bytes + a.other_off[s];
    const u64 k0 = wv_uni((u32)a.step_off[s]), k1 = wv_uni((u32)a.step_off[s + 1]);
    for (u64 k = k0; k < k1; ++k) {
        const u32 op = wv_uni(a.step_op[k]), bl = wv_uni(a.step_blen[k]), ol = wv_uni(a.step_olen[k]);
        int status = 0;
        if (finalized) status = 2;
        else {
            const int rc = op == 0 ? dw_update(wf, a.wf_cap, ed, B, bl, O, ol) : dw_finalize(wf, a.wf_cap, ed, B, bl, O, ol);
            if (rc) status = AVK_ST_CAPACITY;
            else if (op == 1) finalized = true;
        }
        if (wv_lane() == 0) {
            a.step_ed[k] = ed;
            a.step_status[k] = status;
        }
    }
    wv_sync();
    if (a.final_wf)
        for (u32 i = (u32)wv_lane(); i < 2 * ed + 1 && i < a.wf_cap; i += 64) a.final_wf[(u64)s * a.wf_cap + i] = wf[i];
    if (a.final_wf_len && wv_lane() == 0) a.final_wf_len[s] = 2 * ed + 1;
}

namespace lane {

AVK_DEV u32 dwfa_lane_rows() { return 3 * (AVK_DWFA_LANE_W + 1) + 3 * ((2 * AVK_DWFA_LANE_ED + 2 + 3) / 4); }

/* engine 1: this lane works on script s; `lds` = the wave's rows */
AVK_DEV void dwfa_script_lane(const AvkDwfaArgs &a, u32 s, u32 *lds) {
    LCtx c;
    c.p = lds + (u32)wv_lane();
    c.W1 = AVK_DWFA_LANE_W + 1;
    c.ls = 6;
    c.max_nodes = 250;
    c.nm1 = 1;
    c.off_wf = 3 * c.W1;
    c.wfr = (2 * AVK_DWFA_LANE_ED + 2 + 3) / 4;
    c.wfcap = 2 * AVK_DWFA_LANE_ED + 2;
    const u64 k0 = a.step_off[s], k1 = a.step_off[s + 1];
    u32 max_b = 0, max_o = 0;
    for (u64 k = k0; k < k1; ++k) {
        max_b = a.step_blen[k] > max_b ? a.step_blen[k] : max_b;
        max_o = a.step_olen[k] > max_o ? a.step_olen[k] : max_o;
    }
    bool ok = max_b <= 16 * AVK_DWFA_LANE_W && max_o <= 16 * AVK_DWFA_LANE_W;
    for (u32 side = 0; side < 2 && ok; ++side) { /* 2 bits per base into sequence rows 1 (baseline) and 2 (other) */
        const u8 *src = a.bytes + (side ? a.other_off[s] : a.base_off[s]);
        const u32 len = side ? max_o : max_b;
        for (u32 w = 0; w < c.W1; ++w) {
            u32 word = 0;
            for (u32 j = 0; j < 16; ++j) {
                const u32 i = 16 * w + j;
                if (i >= len) break;
                const u8 ch = src[i];
                const u32 code = ch == 'A' ? 0u : (ch == 'C' ? 1u : (ch == 'G' ? 2u : (ch == 'T' ? 3u : 4u)));
                ok = ok && code < 4;
                word |= (code & 3u) << (2 * j);
            }
            c.p[((1 + side) * c.W1 + w) * 64u] = word;
        }
    }
    u32 ed = 0;
    bool finalized = false;
    wf_set(c, 0, 0, 0);
    for (u64 k = k0; k < k1; ++k) {
        int status = 0;
        if (!ok) status = AVK_ST_INVALID_INPUT;
        else if (finalized) status = 2;
        else {
            const int rc = a.step_op[k] == 0 ? dw_update(c, 0, ed, 1, a.step_blen[k], 2, a.step_olen[k], 0xFFFFu)
                                             : dw_finalize(c, 0, ed, 1, a.step_blen[k], 2, a.step_olen[k], 0xFFFFu, c.wfcap);
            if (rc) status = AVK_ST_CAPACITY;
            else if (a.step_op[k] == 1) finalized = true;
        }
        a.step_ed[k] = ed;
        a.step_status[k] = status;
    }
    if (a.final_wf)
        for (u32 i = 0; i < 2 * ed + 1 && i < a.wf_cap; ++i) a.final_wf[(u64)s * a.wf_cap + i] = wf_get(c, 0, i);
    if (a.final_wf_len) a.final_wf_len[s] = 2 * ed + 1;
}

} // namespace lane
} // namespace avk
#endif
