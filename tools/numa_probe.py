"""Why one process in two measures 10 ms per whole-genome call and the other 7: where the process runs and where its pinned pages are, against the H2D / D2H rate it gets.
usage: python tools/numa_probe.py   (several times in a row on one box)"""
import ctypes as C, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aardvark_amd
_libc = C.CDLL("libc.so.6")
def getcpu():
    return int(_libc.sched_getcpu())
cpu0 = getcpu()
allowed = sorted(os.sched_getaffinity(0))
def node_of_cpu(c):
    for p in glob.glob("/sys/devices/system/node/node*/cpu%d" % c):
        return int(p.split("/node/node")[1].split("/")[0])
    return -1
gpu_nodes = []
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    try:
        gpu_nodes.append((p.split("/")[4], int(open(p).read())))
    except Exception:
        pass
ctx = aardvark_amd.Context(0)
n = 128 << 20
h = ctx.host_array((n,), np.uint8)
h[:] = 1
hip = C.CDLL("libamdhip64.so")
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), n) == 0
def rate(kind):
    hip.hipDeviceSynchronize()
    t = time.perf_counter()
    for _ in range(5):
        if kind == 1:
            hip.hipMemcpy(d, h.ctypes.data_as(C.c_void_p), n, 1)
        else:
            hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), d, n, 2)
    hip.hipDeviceSynchronize()
    return 5 * n / (time.perf_counter() - t) / 1e9
r1, r2 = rate(1), rate(2)
# where the pinned pages are: numa_maps line of the mapping that holds h
addr = h.ctypes.data
where = "?"
try:
    for line in open("/proc/self/numa_maps"):
        a = int(line.split()[0], 16)
        if a <= addr < a + n + (64 << 20) and ("N0=" in line or "N1=" in line or "N2=" in line or "N3=" in line):
            best = line
            if a <= addr:
                where = " ".join(x for x in line.split() if x.startswith("N") and "=" in x)
except Exception as e:
    where = "numa_maps: %s" % e
print("cpu at start %d (node %d), now %d (node %d); allowed %d cpus on nodes %s; gpu numa nodes %s; pinned pages %s; H2D %.1f GB/s, D2H %.1f GB/s" % (
    cpu0, node_of_cpu(cpu0), getcpu(), node_of_cpu(getcpu()), len(allowed), sorted(set(node_of_cpu(c) for c in allowed)), gpu_nodes[:3], where, r1, r2), flush=True)
os._exit(0)
