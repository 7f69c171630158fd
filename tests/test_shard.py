"""The multi-GPU rule of the product — shard = hash(region_id) % ranks — stated once in C (include/aardvark_amd.h: avk_region_hash / avk_region_shard, used by
avk_packed_shard_make and the command-line tool) and once in Python (aardvark_amd/dist.py): the two must agree; and a packed batch cut into its shards, solved shard
by shard and scattered back is the batch solved whole (host-side functions: no GPU needed to cut and scatter)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import aardvark_amd
import oracle_lib
import scenarios
from aardvark_amd import CompactBatch, PackedBatch, ResultBatch, dist, synth
from aardvark_amd._abi import AvkPackedBatch, AvkResultBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_python_and_the_c_statement_of_the_hash_agree(tmp_path):
    src = tmp_path / "h.c"
    src.write_text('#include <stdio.h>\n#include "aardvark_amd.h"\nint main(void) { unsigned long long x; while (scanf("%llu", &x) == 1) printf("%llu %u %u\\n", '
                   '(unsigned long long)avk_region_hash(x), avk_region_shard(x, 8), avk_region_shard(x, 3)); return 0; }\n')
    exe = tmp_path / "h"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    ids = np.concatenate([np.arange(0, 2000, dtype=np.uint64), np.random.default_rng(1).integers(0, 2**63, size=2000, dtype=np.uint64), np.array([2**64 - 1], np.uint64)])
    out = subprocess.run([str(exe)], input="\n".join(str(int(x)) for x in ids), capture_output=True, text=True, check=True).stdout.split()
    got = np.array(out, dtype=np.uint64).reshape(-1, 3)
    h = dist.region_hash(ids)
    assert np.array_equal(got[:, 0], h)
    assert np.array_equal(got[:, 1], h % np.uint64(8)) and np.array_equal(got[:, 2], h % np.uint64(3))
    for world in (2, 3, 8):
        parts = [dist.shard_indices(ids, r, world) for r in range(world)]
        assert sorted(np.concatenate(parts).tolist()) == list(range(ids.size))


def shard_api():
    lib = aardvark_amd.load_library()
    lib.avk_packed_shard_make.argtypes = [C.POINTER(AvkPackedBatch), C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.avk_packed_shard_batch.restype = C.POINTER(AvkPackedBatch)
    lib.avk_packed_shard_batch.argtypes = [C.c_void_p]
    lib.avk_packed_shard_regions.restype = C.c_uint64
    lib.avk_packed_shard_regions.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64))]
    lib.avk_packed_shard_scatter.argtypes = [C.c_void_p, C.POINTER(AvkResultBatch), C.POINTER(AvkResultBatch)]
    lib.avk_packed_shard_free.argtypes = [C.c_void_p]
    return lib


def shard_as_python(lib, handle):
    """the shard's packed batch as a PackedBatch (copies) and its regions' indices in the whole batch"""
    b = lib.avk_packed_shard_batch(handle).contents
    n, nv, na = int(b.n_regions), int(b.n_variants), int(b.allele_bytes_len)
    take = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt).copy() if p else None
    pb = PackedBatch(contig_idx=take(b.contig_idx, n, np.uint16), start=take(b.start, n, np.uint32), len=take(b.len, n, np.uint16), t_cnt=take(b.t_cnt, n, np.uint8),
                     q_cnt=take(b.q_cnt, n, np.uint8), var_rel_pos=take(b.var_rel_pos, nv, np.uint16), var_type_zyg=take(b.var_type_zyg, nv, np.uint8),
                     a0_len=take(b.a0_len, nv, np.uint8), a1_len=take(b.a1_len, nv, np.uint8), var_raw_space=take(b.var_raw_space, nv, np.uint32),
                     allele_bytes=take(b.allele_bytes, na, np.uint8))
    idx = C.POINTER(C.c_uint64)()
    m = lib.avk_packed_shard_regions(handle, C.byref(idx))
    return pb, np.ctypeslib.as_array(idx, shape=(max(int(m), 1),))[:int(m)].copy()


@pytest.mark.parametrize("world", [2, 3])
def test_shards_of_a_packed_batch_hold_the_regions_the_python_rule_names(world):
    lib = shard_api()
    contig, batch = synth.config_indel_mix_v2(n_truth=3000, contig_len=1_200_000)
    whole = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
    cb = whole.c_struct()
    ids = np.ascontiguousarray(batch.region_id + np.uint64(77), np.uint64)
    seen = []
    for rank in range(world):
        h = C.c_void_p()
        assert lib.avk_packed_shard_make(C.byref(cb), ids.ctypes.data_as(C.POINTER(C.c_uint64)), 0, rank, world, C.byref(h)) == 0
        pb, idx = shard_as_python(lib, h)
        assert np.array_equal(idx, dist.shard_indices(ids, rank, world))
        # the shard is what gathering these regions by hand gives
        voff = np.concatenate([[0], np.cumsum(whole.t_cnt.astype(np.int64) + whole.q_cnt)])
        assert np.array_equal(pb.start, whole.start[idx]) and np.array_equal(pb.t_cnt, whole.t_cnt[idx]) and np.array_equal(pb.len, whole.len[idx])
        calls = np.concatenate([np.arange(voff[r], voff[r + 1]) for r in idx]) if idx.size else np.zeros(0, np.int64)
        assert np.array_equal(pb.var_rel_pos, whole.var_rel_pos[calls]) and np.array_equal(pb.var_type_zyg, whole.var_type_zyg[calls])
        assert int(pb.allele_bytes.size) == int((whole.a0_len[calls].astype(np.int64) + whole.a1_len[calls]).sum())
        lib.avk_packed_shard_free(h)
        seen.append(idx)
    assert sorted(np.concatenate(seen).tolist()) == list(range(batch.n_regions))


@pytest.mark.gpu
def test_shards_solved_one_by_one_and_scattered_are_the_batch_solved_whole(oracle):
    lib = shard_api()
    contigs, batch = synth.config_genome(scale=0.01)
    whole = PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
    cb = whole.c_struct()
    ctx = aardvark_amd.Context(0)
    try:
        ctx.upload_reference(contigs)
        want = ctx.solve_packed(whole, res=ResultBatch(whole, sequences=False, group_metrics=False))
        ref = oracle_lib.compare_batch(oracle, batch, contigs, threads=8, group_metrics=False)
        assert want.diff(ref) == []
        for world in (2, 5):
            got = ResultBatch(whole, sequences=False, group_metrics=False)
            tally = np.zeros_like(got.tally)
            for rank in range(world):
                h = C.c_void_p()
                assert lib.avk_packed_shard_make(C.byref(cb), batch.region_id.ctypes.data_as(C.POINTER(C.c_uint64)), 0, rank, world, C.byref(h)) == 0
                pb, idx = shard_as_python(lib, h)
                res = ctx.solve_packed(pb, res=ResultBatch(pb, sequences=False, group_metrics=False))
                sr, gr = res.c_struct(), got.c_struct()
                assert lib.avk_packed_shard_scatter(h, C.byref(sr), C.byref(gr)) == 0
                tally += res.tally
                lib.avk_packed_shard_free(h)
            got.tally[:] = tally
            assert got.diff(want) == []
    finally:
        ctx.close()


@pytest.mark.gpu
def test_tally_all_reduce_over_rccl_at_world_size_one():
    """avk_tally_allreduce with a real RCCL communicator (one rank: the pool has one GPU): the sums come back as they went in, through ncclAllReduce on the context's stream"""
    rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    lib = aardvark_amd.load_library()
    lib.avk_tally_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    ctx = aardvark_amd.Context(0)
    try:
        tally = np.arange(aardvark_amd.TALLY_LEN, dtype=np.uint64) * np.uint64(1_000_003) + np.uint64(2**40)
        keep = tally.copy()
        assert lib.avk_tally_allreduce(ctx.handle, comm, tally.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        assert np.array_equal(tally, keep)
    finally:
        ctx.close()
        rccl.ncclCommDestroy(comm)
