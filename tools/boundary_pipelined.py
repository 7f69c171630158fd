"""One process, whole genomes back to back through avk_compare_packed_submit / avk_wait (two in flight, pinned arrays): the command rocprofv3 traces for
profiles/r05_boundary_timeline.txt.  usage: python tools/boundary_pipelined.py [scale] [genomes] [opt=value,...] [in flight]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompactBatch, PackedBatch

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
genomes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
contigs, batch = synth.config_genome(scale=scale)
ctx = aardvark_amd.Context(0)
ctx.set_option("emit_group_metrics", 0)
for kv in (sys.argv[3] if len(sys.argv) > 3 else "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.upload_reference(contigs)
whole = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 2
sets = [(ctx.pinned_packed(whole), ctx.pinned_results(whole, packed="only")) for _ in range(depth)]
for hb, res in sets:
    ctx.solve_packed(hb, res=res)
tickets = [ctx.submit_packed(sets[k][0], res=sets[k][1]) for k in range(depth - 1)]
t0 = time.perf_counter()
for k in range(depth - 1, genomes + depth - 1):
    tickets.append(ctx.submit_packed(sets[k % depth][0], res=sets[k % depth][1]))
    tickets.pop(0).wait()
for t in tickets:
    t.wait()
print("%d genomes back to back, %d in flight: %.3f ms per genome" % (genomes, depth, (time.perf_counter() - t0) / genomes * 1e3), flush=True)
