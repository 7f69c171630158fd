"""Does a host-to-device copy run BESIDE kernels on this box?  The asynchronous boundary (avk_compare_packed_submit) rests on it: copies of batch k + 1 under the solve of
batch k.  Prints the box's name, the time of a 96 MB pinned copy alone and while whole-genome resident steps are queued, and the pipelined rate.
usage on the GPU box: python tools/gpu_overlap_probe.py"""
import ctypes as C, os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig, CompactBatch, PackedBatch
aardvark_amd.load_library()
hip = C.CDLL(next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
print("box %s, HSA_ENABLE_SDMA=%s, kernel %s" % (socket.gethostname(), os.environ.get("HSA_ENABLE_SDMA"), os.uname().release))
contigs, batch = synth.config_genome(scale=1.0)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
nbytes = 96 << 20
host = ctx.host_array((nbytes,), np.uint8)
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), nbytes) == 0
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
def copy_ms():
    t0 = time.perf_counter()
    assert hip.hipMemcpyAsync(dev, host.ctypes.data, nbytes, 1, st) == 0
    assert hip.hipStreamSynchronize(st) == 0
    return (time.perf_counter() - t0) * 1e3
copy_ms()
alone = min(copy_ms() for _ in range(5))
rb = ctx.upload(batch)
cfg = CompareConfig(enable_sequences=False)
for _ in range(3):
    ctx.compare_resident(rb, cfg)
ctx.synchronize()
for _ in range(12):
    ctx.compare_resident(rb, cfg)  # ~30 ms of queued solver launches
beside = [copy_ms() for _ in range(3)]
ctx.synchronize()
print("96 MB pinned host-to-device copy: alone %.2f ms (%.1f GB/s); while resident steps are running %s ms" % (alone, nbytes / alone / 1e6, " ".join("%.2f" % x for x in beside)))
whole = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
sets = [(ctx.pinned_packed(whole), ctx.pinned_results(whole, packed="only")) for _ in range(2)]
for hb, res in sets:
    ctx.solve_packed(hb, res=res)
t0 = time.perf_counter()
for _ in range(10):
    ctx.solve_packed(sets[0][0], res=sets[0][1])
sync_ms = (time.perf_counter() - t0) / 10 * 1e3
tk = ctx.submit_packed(sets[0][0], res=sets[0][1])
for k in range(1, 5):  # (warm-up in the timed pattern: the second batch in flight takes a second set of device buffers)
    nx = ctx.submit_packed(sets[k & 1][0], res=sets[k & 1][1]); tk.wait(); tk = nx
t0 = time.perf_counter()
for k in range(1, 21):
    nx = ctx.submit_packed(sets[k & 1][0], res=sets[k & 1][1]); tk.wait(); tk = nx
tk.wait()
print("whole genome: synchronous call %.2f ms, two in flight %.2f ms per genome" % (sync_ms, (time.perf_counter() - t0) / 20 * 1e3))
