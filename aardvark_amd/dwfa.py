"""ctypes wrapper of avk_dwfa_script_batch: the device aligner driven by scripts of DWFALite calls (reference src/dwfa/dynamic_wfa.rs:23-276).
A script = (baseline bytes, other bytes, [(op, baseline_prefix_len, other_prefix_len), ...]) with op "update" or "finalize"."""
import ctypes as C

import numpy as np

u8p, u32p, u64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
ARGTYPES = [C.c_int, C.c_uint32, u8p, C.c_uint64, u64p, u64p, u64p, u8p, u32p, u32p, u32p, i32p, C.c_uint32, u32p, u32p]


def pack_scripts(scripts):
    """-> the flat arrays of the C entry"""
    blob = bytearray()
    base_off, other_off, step_off, op, bl, ol = [], [], [0], [], [], []
    for b, o, steps in scripts:
        base_off.append(len(blob)); blob += bytes(b)
        other_off.append(len(blob)); blob += bytes(o)
        for kind, nb, no in steps:
            op.append(0 if kind == "update" else 1); bl.append(nb); ol.append(no)
        step_off.append(len(op))
    blob += b"\0" * 16
    a = lambda x, dt: np.ascontiguousarray(np.array(x if len(x) else [0], dtype=dt))
    return (np.frombuffer(bytes(blob), np.uint8).copy(), a(base_off, np.uint64), a(other_off, np.uint64), np.array(step_off, np.uint64), a(op, np.uint8),
            a(bl, np.uint32), a(ol, np.uint32), len(scripts), len(op))


def run_scripts(fn, prefix_args, scripts, engine, wf_cap=1 << 15):
    """fn = avk_dwfa_script_batch (prefix_args = [ctx handle]) or the emulator's emu_dwfa_script_batch (prefix_args = []).
    Returns (ed per step, status per step, final wavefronts as lists), steps grouped per script."""
    blob, base_off, other_off, step_off, op, bl, ol, n, n_steps = pack_scripts(scripts)
    ed = np.zeros(max(n_steps, 1), np.uint32)
    st = np.full(max(n_steps, 1), -1, np.int32)
    wf = np.zeros((max(n, 1), wf_cap), np.uint32)
    wfl = np.zeros(max(n, 1), np.uint32)
    p = lambda arr, t: arr.ctypes.data_as(t)
    rc = fn(*prefix_args, engine, n, p(blob, u8p), blob.size, p(base_off, u64p), p(other_off, u64p), p(step_off, u64p), p(op, u8p), p(bl, u32p), p(ol, u32p),
            p(ed, u32p), p(st, i32p), wf_cap, p(wf, u32p), p(wfl, u32p))
    if rc != 0:
        raise RuntimeError("dwfa script batch failed: %d" % rc)
    out = []
    for s in range(n):
        lo, hi = int(step_off[s]), int(step_off[s + 1])
        out.append((ed[lo:hi].tolist(), st[lo:hi].tolist(), wf[s, :int(wfl[s])].tolist()))
    return out


def device_scripts(ctx, scripts, engine, wf_cap=1 << 15):
    """avk_dwfa_script_batch on the GPU of `ctx` (aardvark_amd.Context)"""
    fn = ctx.lib.avk_dwfa_script_batch
    fn.argtypes = [C.c_void_p] + ARGTYPES
    return run_scripts(fn, [ctx.handle], scripts, engine, wf_cap)
