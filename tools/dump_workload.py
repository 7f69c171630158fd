"""Writes the benchmark workload (aardvark_amd/synth.py::config_genome) as one flat file for the C++ harnesses (tools/first_step_probe.cpp).
usage: python tools/dump_workload.py <scale> <out file> [gap=50] [dense=0] [shards=1]   (shards = K: rank 0's hash shard of K, aardvark_amd/dist.py)
layout (little endian): u64 magic 'AVKWORK1', u64 n_contigs, u64 len[n_contigs], contig bytes (each padded to 16), u64 n_regions, u64 n_variants,
u64 allele_bytes_len, then the arrays of avk_region_batch in declaration order, each padded to 16 bytes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aardvark_amd import synth

scale = float(sys.argv[1])
out = sys.argv[2]
gap = int(sys.argv[3]) if len(sys.argv) > 3 else 50
kw = dict(close_frac=0.10, str_frac=0.15, multi_frac=0.05) if len(sys.argv) > 4 and sys.argv[4] == "1" else {}
contigs, b = synth.config_genome(scale=scale, threads=8, gap=gap, **kw)
if len(sys.argv) > 5 and int(sys.argv[5]) > 1:
    from aardvark_amd import dist as avk_dist
    b = avk_dist.gather_calls(avk_dist.shard_batch(b, 0, int(sys.argv[5])))


def pad(f):
    n = (-f.tell()) % 16
    if n:
        f.write(b"\0" * n)


with open(out, "wb") as f:
    f.write(b"AVKWORK1")
    f.write(np.array([len(contigs)], np.uint64).tobytes())
    f.write(np.array([c.size for c in contigs], np.uint64).tobytes())
    pad(f)
    for c in contigs:
        f.write(np.ascontiguousarray(c, np.uint8).tobytes())
        pad(f)
    f.write(np.array([b.n_regions, b.n_variants, b.allele_bytes.size], np.uint64).tobytes())
    pad(f)
    for name in ("region_id", "contig_idx", "start", "end", "t_off", "t_cnt", "q_off", "q_cnt", "var_pos", "var_type", "var_zyg", "var_raw_space",
                 "a0_off", "a0_len", "a1_off", "a1_len", "allele_bytes"):
        f.write(getattr(b, name).tobytes())
        pad(f)
print("wrote %s: %d contigs, %d regions, %d calls, %.1f MB" % (out, len(contigs), b.n_regions, b.n_variants, os.path.getsize(out) / 1e6))
