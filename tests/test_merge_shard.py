"""Merge on several GPUs (SURVEY.md 8e "merge (config 5): same sharding", BASELINE configs[4]): the reference maps merge regions exactly like compare regions
(src/main.rs:463-478) and the only state it keeps across them is MergeSummaryWriter's (reason, type, input) -> (pass, fail) map (src/writers/merge_summary.rs:12-18).
So: a packed multi-region batch is cut by the ONE rule shard = hash(region_id) % ranks (avk_packed_multi_shard_make), every rank solves its shard, the per-region
results are scattered back for the writers, and the ranks' summary counters — a dense block of sums (avk_merge_counts) — are added up by one all-reduce (RCCL on
the GPU box, gloo here) and written as the table (avf_write_merge_summary_counts), byte for byte the table of the unsharded job."""
import ctypes as C
import gzip
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import merge_oracle as mo  # noqa: E402
import oracle_lib  # noqa: E402
import aardvark_amd  # noqa: E402
from aardvark_amd import dist as avk_dist, feeder, synth  # noqa: E402
from aardvark_amd.merge import (MergeResult, PackedMultiBatch, merge_counts, merge_counts_len, shard_packed_multi)  # noqa: E402


def small_job(scale=0.0008):
    contigs, mb = synth.config_genome_merge(scale=scale, k=3, threads=4)
    return contigs, mb, PackedMultiBatch.from_multi(mb)


def random_results(n, k, seed, unsolved=0.02):
    """consistent (status, classification, members) triples of every kind, some regions unsolved"""
    rng = np.random.default_rng(seed)
    cls = rng.integers(0, 5, n).astype(np.uint8)
    masks = rng.integers(1, 1 << k, n).astype(np.uint64)
    index = rng.integers(0, k, n).astype(np.uint64)
    members = np.where((cls == 2) | (cls == 3), masks, np.where(cls == 4, index, 0)).astype(np.uint64)
    status = np.where(rng.random(n) < unsolved, 3, 0).astype(np.int32)
    return MergeResult(status, cls, members, k)


def take(res, idx):
    return MergeResult(res.status[idx], res.classification[idx], res.members[idx], res.n_inputs)


def oracle_merge(oracle, mb, contigs, threads=4):
    """solve_merge_region with the majority strategy for a MultiBatch of three inputs: pairs from the C oracle, the decision from the restated rule"""
    import bench
    pb = bench.pair_batch_of(mb)
    st, ex = oracle_lib.optimize_pairs(oracle, pb, contigs, 50, threads=threads)
    s, c, m = mo.classify_k3_majority(st, ex)
    return MergeResult(s, c, m, 3)


@pytest.mark.parametrize("world", [2, 3])
def test_shards_of_a_packed_multi_batch_hold_the_regions_the_python_rule_names(world):
    lib = aardvark_amd.load_library()
    _, mb, whole = small_job()
    ids = np.ascontiguousarray(mb.region_id + np.uint64(5), np.uint64)
    k = whole.n_inputs
    per_region = whole.in_cnt.astype(np.int64).reshape(-1, k).sum(axis=1)
    voff = np.concatenate([[0], np.cumsum(per_region)])
    alen = whole.a0_len.astype(np.int64) + whole.a1_len
    aoff = np.concatenate([[0], np.cumsum(alen)])
    seen = []
    for rank in range(world):
        shard, idx = shard_packed_multi(lib, whole, ids, rank, world)
        assert np.array_equal(idx, avk_dist.shard_indices(ids, rank, world))
        assert shard.n_inputs == k and shard.n_regions == idx.size
        assert np.array_equal(shard.start, whole.start[idx]) and np.array_equal(shard.len, whole.len[idx]) and np.array_equal(shard.contig_idx, whole.contig_idx[idx])
        assert np.array_equal(shard.in_cnt.reshape(-1, k), whole.in_cnt.reshape(-1, k)[idx])
        calls = np.concatenate([np.arange(voff[r], voff[r + 1]) for r in idx]) if idx.size else np.zeros(0, np.int64)
        for f in ("var_rel_pos", "var_type_zyg", "a0_len", "a1_len"):
            assert np.array_equal(getattr(shard, f), getattr(whole, f)[calls]), f
        want_bytes = np.concatenate([whole.allele_bytes[aoff[v]:aoff[v + 1]] for v in calls]) if calls.size else np.zeros(0, np.uint8)
        assert np.array_equal(shard.allele_bytes[:want_bytes.size], want_bytes)
        # the shard widens to the regions the whole batch widens to
        a, b = shard.widen().regions(), whole.widen().regions()
        for j, r in enumerate(idx[:50]):
            assert a[j]["inputs"] == b[int(r)]["inputs"] and (a[j]["start"], a[j]["end"]) == (b[int(r)]["start"], b[int(r)]["end"])
        seen.append(idx)
    assert sorted(np.concatenate(seen).tolist()) == list(range(mb.n_regions))


def test_scatter_puts_a_shards_results_at_its_regions():
    lib = aardvark_amd.load_library()
    from aardvark_amd.merge import AvkPackedMultiBatch, _shard_api
    _shard_api(lib)
    _, mb, whole = small_job()
    res = random_results(mb.n_regions, 3, 11)
    got = MergeResult(np.full(mb.n_regions, -1, np.int32), np.full(mb.n_regions, 9, np.uint8), np.zeros(mb.n_regions, np.uint64), 3)
    cb = whole.c_struct()
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for rank in range(3):
        h = C.c_void_p()
        assert lib.avk_packed_multi_shard_make(C.byref(cb), P(mb.region_id, C.c_uint64), 0, rank, 3, C.byref(h)) == 0
        idx = avk_dist.shard_indices(mb.region_id, rank, 3)
        part = take(res, idx)
        st, cl, me = np.ascontiguousarray(part.status), np.ascontiguousarray(part.classification), np.ascontiguousarray(part.members)
        assert lib.avk_packed_multi_shard_scatter(h, P(st, C.c_int32), P(cl, C.c_uint8), P(me, C.c_uint64), P(got.status, C.c_int32), P(got.classification, C.c_uint8),
                                                  P(got.members, C.c_uint64)) == 0
        lib.avk_packed_multi_shard_free(h)
    assert np.array_equal(got.status, res.status) and np.array_equal(got.classification, res.classification) and np.array_equal(got.members, res.members)
    # region_id NULL: ids first_id + r
    h = C.c_void_p()
    assert lib.avk_packed_multi_shard_make(C.byref(cb), None, 1000, 1, 2, C.byref(h)) == 0
    idx = C.POINTER(C.c_uint64)()
    m = int(lib.avk_packed_multi_shard_regions(h, C.byref(idx)))
    assert np.array_equal(np.ctypeslib.as_array(idx, shape=(m,)), avk_dist.shard_indices(np.arange(mb.n_regions, dtype=np.uint64) + np.uint64(1000), 1, 2))
    lib.avk_packed_multi_shard_free(h)
    assert lib.avk_packed_multi_shard_make(C.byref(cb), None, 0, 2, 2, C.byref(h)) != 0  # rank outside the world


def test_the_shards_counters_add_up_to_the_jobs_and_write_the_same_summary(tmp_path):
    lib = aardvark_amd.load_library()
    _, mb, whole = small_job()
    k = 3
    assert merge_counts_len(lib, k) == (2 + 2 * 8 + 3) * 12 * 3 * 2 and merge_counts_len(lib, 11) == 0 and merge_counts_len(lib, 1) == 0
    res = random_results(mb.n_regions, k, 7)
    job = merge_counts(lib, whole, res)
    # every call of a solved region is counted once, as a pass or a fail
    per_region = whole.in_cnt.astype(np.int64).reshape(-1, k).sum(axis=1)
    assert int(job.sum()) == int(per_region[res.status == 0].sum())
    for world in (2, 5):
        total = np.zeros_like(job)
        for rank in range(world):
            shard, idx = shard_packed_multi(lib, whole, mb.region_id, rank, world)
            merge_counts(lib, shard, take(res, idx), total)  # ADDS
        assert np.array_equal(total, job)
    tags = ["hifi", "ont", "ilmn"]
    a, b = str(tmp_path / "arrays.tsv"), str(tmp_path / "counts.tsv")
    feeder.write_merge_summary(a, mb, res, tags)
    feeder.write_merge_summary_counts(b, k, job, tags)
    text = open(a).read()
    assert text == open(b).read() and text.count("\n") > 20
    rows = [l.split("\t")[0] for l in text.splitlines()[1:]]
    assert any(r.startswith("no_conflict_0_2") for r in rows) and any(r.startswith("conflict_select_1") for r in rows) and "identical" in rows and "different" in rows
    feeder.write_merge_summary_counts(str(tmp_path / "c.csv"), k, job, tags)
    feeder.write_merge_summary(str(tmp_path / "a.csv"), mb, res, tags)
    assert open(tmp_path / "c.csv").read() == open(tmp_path / "a.csv").read()
    # nothing solved: the empty file of the reference's csv writer
    none = MergeResult(np.full(mb.n_regions, 3, np.int32), res.classification, res.members, k)
    feeder.write_merge_summary_counts(b, k, merge_counts(lib, whole, none), tags)
    assert open(b).read() == ""
    with pytest.raises(feeder.FeederError):
        feeder.write_merge_summary_counts(b, k, job[:-2], tags)
    with pytest.raises(ValueError):  # a selected index outside the inputs
        merge_counts(lib, whole, MergeResult(np.zeros(mb.n_regions, np.int32), np.full(mb.n_regions, 4, np.uint8), np.full(mb.n_regions, 3, np.uint64), k))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = aardvark_amd.load_library()
    oracle = oracle_lib.load()
    contigs, mb, whole = small_job()
    shard, idx = shard_packed_multi(lib, whole, mb.region_id, rank, world)  # what bench.py --gpus N hands to each rank's avk_merge_packed
    res = oracle_merge(oracle, shard.widen(), contigs, threads=2)
    counts = torch.from_numpy(merge_counts(lib, shard, res).view(np.int64).copy())
    avk_dist.allreduce_counts(counts)
    np.save(os.path.join(out_dir, "counts_%d.npy" % rank), counts.numpy().view(np.uint64))
    np.save(os.path.join(out_dir, "idx_%d.npy" % rank), idx)
    np.save(os.path.join(out_dir, "res_%d.npy" % rank), np.stack([res.status.astype(np.int64), res.classification.astype(np.int64), res.members.astype(np.int64)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_a_merge_and_allreduce_its_summary_counters(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    lib = aardvark_amd.load_library()
    oracle = oracle_lib.load()
    contigs, mb, whole = small_job()
    want = oracle_merge(oracle, mb, contigs)
    assert (want.classification == 3).sum() > 10 and (want.classification == 1).sum() > 10
    job = merge_counts(lib, whole, want)
    c0, c1 = np.load(tmp_path / "counts_0.npy"), np.load(tmp_path / "counts_1.npy")
    assert np.array_equal(c0, c1) and np.array_equal(c0, job)
    got = np.full((3, mb.n_regions), -1, np.int64)
    for r in range(world):
        idx = np.load(tmp_path / ("idx_%d.npy" % r))
        assert (got[0, idx] == -1).all()
        got[:, idx] = np.load(tmp_path / ("res_%d.npy" % r))
    assert np.array_equal(got[0], want.status) and np.array_equal(got[1], want.classification) and np.array_equal(got[2].astype(np.uint64), want.members)
    a, b = str(tmp_path / "whole.tsv"), str(tmp_path / "ranks.tsv")
    feeder.write_merge_summary(a, mb, want)
    feeder.write_merge_summary_counts(b, 3, c0)
    assert open(a).read() == open(b).read() != ""


def merge_cli():
    return os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_merge")


@pytest.mark.gpu
def test_merge_tool_sharded_over_two_contexts_writes_the_files_of_one(tmp_path):
    """aardvark_amd_merge --devices 0,0 (two contexts, the job cut by avk_region_shard, counters summed on the host because the entries repeat) against the
    single-context run: passing.vcf.gz, both BED files and the summary byte for byte"""
    from test_merge_outputs import write_case
    p, _ = write_case(tmp_path, 1500, 600_000)
    runs = {}
    for name, extra in (("one", []), ("two", ["--devices", "0,0", "-v"]), ("three_skip", ["--devices", "0,0,0", "--skip", "7", "--take", "300"]), ("one_skip", ["--skip", "7", "--take", "300"])):
        out, summary = str(tmp_path / ("out_" + name)), str(tmp_path / (name + ".tsv"))
        cmd = [merge_cli(), "-r", p["fa"]] + [x for v in p["vcfs"] for x in ("-i", v)] + ["-b", p["bed"], "-o", out, "--output-summary", summary, "--merge-strategy", "all",
                                                                                           "--conflict-select", "1"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        runs[name] = (out, summary, r.stderr)
    assert "2 contexts, regions sharded by hash(region_id) % 2" in runs["two"][2]
    strip = lambda x: b"\n".join(l for l in x.split(b"\n") if not l.startswith(b"##aardvark_command"))
    for a, b in (("one", "two"), ("one_skip", "three_skip")):
        for name in ("passing.vcf.gz", "regions.bed.gz", "failed_regions.bed.gz"):
            x, y = (gzip.open(os.path.join(runs[n][0], name), "rb").read() for n in (a, b))
            assert strip(x) == strip(y) and (len(x) > 100 or name == "failed_regions.bed.gz"), name  # (this strategy fails no region)
        assert open(runs[a][1]).read() == open(runs[b][1]).read() != ""
        solved = [l for l in runs[a][2].splitlines() if l.startswith("Solved:error")]
        assert solved and solved == [l for l in runs[b][2].splitlines() if l.startswith("Solved:error")]


@pytest.mark.gpu
def test_counts_all_reduce_over_rccl_at_world_size_one():
    """avk_counts_allreduce with a real RCCL communicator (one rank: the pool has one GPU) on a merge's block of summary counters"""
    from aardvark_amd.merge import _shard_api
    rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    lib = _shard_api(aardvark_amd.load_library())
    ctx = aardvark_amd.Context(0)
    try:
        n = merge_counts_len(lib, 3)
        counts = np.arange(n, dtype=np.uint64) * np.uint64(1_000_003) + np.uint64(2 ** 41)
        keep = counts.copy()
        assert lib.avk_counts_allreduce(ctx.handle, comm, counts.ctypes.data_as(C.POINTER(C.c_uint64)), n) == 0
        assert np.array_equal(counts, keep)
    finally:
        ctx.close()
        rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
def test_shards_merged_one_by_one_and_scattered_are_the_batch_merged_whole():
    from aardvark_amd.merge import MergeConfig, merge_multi_batch
    lib = aardvark_amd.load_library()
    contigs, mb, whole = small_job(0.01)
    ctx = aardvark_amd.Context(0)
    try:
        ctx.upload_reference(contigs)
        cfg = MergeConfig(majority_voting_enabled=True)
        want = merge_multi_batch(ctx, whole, cfg)
        job = merge_counts(lib, whole, want)
        for world in (2, 8):
            got = MergeResult(np.full(mb.n_regions, -1, np.int32), np.zeros(mb.n_regions, np.uint8), np.zeros(mb.n_regions, np.uint64), 3)
            total = np.zeros_like(job)
            for rank in range(world):
                shard, idx = shard_packed_multi(lib, whole, mb.region_id, rank, world)
                res = merge_multi_batch(ctx, shard, cfg)
                got.status[idx], got.classification[idx], got.members[idx] = res.status, res.classification, res.members
                merge_counts(lib, shard, res, total)
            assert np.array_equal(got.status, want.status) and np.array_equal(got.classification, want.classification) and np.array_equal(got.members, want.members)
            assert np.array_equal(total, job)
    finally:
        ctx.close()
