"""Who is in the slowest narrow tiles of the lane kernel: one resident whole-genome step with the profiling build
(make -C aardvark_amd/csrc lane-slow-tiles), whose kernel prints a line per lane of every narrow tile over the tick limit.
usage on the GPU box: AVK_LIB=libaardvark_amd_slowtiles.so python tools/gpu_slow_tiles.py [scale] > gpurun_out/slowtiles.txt
then, anywhere: python tools/gpu_slow_tiles.py --describe gpurun_out/slowtiles.txt [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aardvark_amd import synth

if len(sys.argv) > 1 and sys.argv[1] == "--describe":
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    tiles, busy = [], []
    for line in open(sys.argv[2]):
        f = line.split()
        if line.startswith("tile "):
            tiles.append((int(f[6]), int(f[2]), int(f[4])))
        elif line.startswith("busylane"):
            busy.append(dict(nm=int(f[2]), claim=int(f[4]), lane=int(f[6]), orig=int(f[8]), pops=int(f[10]), diag=int(f[12]), words=int(f[14])))
    tiles.sort(reverse=True)
    print("%d narrow tiles; ticks of the slowest 12 (nm = masks per side of the class: 2 one call, 4 two calls, 8 three calls):" % len(tiles))
    for t in tiles[:12]:
        print("   nm %d claim %6d ticks %d" % (t[1], t[2], t[0]))
    import statistics
    for nm in (2, 4, 8):
        tt = [t[0] for t in tiles if t[1] == nm]
        if tt:
            print("   nm %d: %d tiles, median %d, 90 %% %d, max %d ticks" % (nm, len(tt), statistics.median(tt), sorted(tt)[int(0.9 * len(tt))], max(tt)))
    contigs, b = synth.config_genome(scale=scale)
    slow = {(t[1], t[2]) for t in tiles[:200]}
    busy.sort(key=lambda r: -(r["diag"] + 20 * r["pops"]))
    print("%d busy lanes; the 40 busiest (pops, diagonals extended, words compared; * = in one of the 200 slowest tiles):" % len(busy))
    for r in busy[:40]:
        o = r["orig"]
        def side(off, cnt):
            return [(int(b.var_pos[int(off[o]) + k] - b.start[o]), int(b.a0_len[int(off[o]) + k]), int(b.a1_len[int(off[o]) + k]), int(b.var_type[int(off[o]) + k]), int(b.var_zyg[int(off[o]) + k])) for k in range(int(cnt[o]))]
        print("%s nm %d claim %6d lane %2d pops %4d diagonals %6d words %6d  L %3d truth %s query %s" % (
            "*" if (r["nm"], r["claim"]) in slow else " ", r["nm"], r["claim"], r["lane"], r["pops"], r["diag"], r["words"], int(b.end[o] - b.start[o]), side(b.t_off, b.t_cnt), side(b.q_off, b.q_cnt)))
    sys.exit(0)

assert "slowtiles" in os.environ.get("AVK_LIB", ""), "set AVK_LIB=libaardvark_amd_slowtiles.so"
import aardvark_amd
from aardvark_amd import CompareConfig
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
ctx.upload_reference(contigs)
rb = ctx.upload(batch)
ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
ctx.synchronize()
