"""Multi-GPU path on CPU: two gloo ranks shard the regions by region_id hash, solve their shards
(kernel logic through the lane emulator), sum the tally blocks with an all-reduce and must get the
oracle's whole-batch tally; per-variant decisions gathered from the shards must equal the
single-process run.  bench.py uses the same `shard_of` / all-reduce on RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import dist as avk_dist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    contigs, batch = scenarios.chr20_small(1200)
    mine = avk_dist.shard_indices(batch.region_id, rank, world)
    shard = avk_dist.shard_batch(batch, rank, world)  # what bench.py --scaling strong hands to each rank
    assert np.array_equal(shard.region_id, batch.region_id[mine])
    res = emu_lib.compare_batch(shard, contigs, n_waves=4, threads=2)
    tally = torch.from_numpy(res.tally.astype(np.int64))
    avk_dist.allreduce_tally(tally)
    np.save(os.path.join(out_dir, "tally_%d.npy" % rank), tally.numpy())
    np.save(os.path.join(out_dir, "idx_%d.npy" % rank), mine)
    np.save(os.path.join(out_dir, "status_%d.npy" % rank), res.status)
    np.save(os.path.join(out_dir, "gm_%d.npy" % rank), res.group_metrics)
    # what bench.py --scaling strong does: per-shard checksums summed over the ranks (int64 wrap-around = sum modulo 2^64)
    chk = torch.from_numpy(np.array([avk_dist.result_checksum(shard, res)], np.uint64).view(np.int64).copy())
    dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    np.save(os.path.join(out_dir, "chk_%d.npy" % rank), chk.numpy().view(np.uint64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_tally_allreduce(tmp_path, oracle):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    contigs, batch = scenarios.chr20_small(1200)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    t0 = np.load(tmp_path / "tally_0.npy")
    t1 = np.load(tmp_path / "tally_1.npy")
    assert np.array_equal(t0, t1)
    assert np.array_equal(t0.astype(np.uint64), want.tally)
    seen = np.zeros(batch.n_regions, bool)
    for r in range(world):
        idx = np.load(tmp_path / ("idx_%d.npy" % r))
        assert not seen[idx].any()
        seen[idx] = True
        assert np.array_equal(np.load(tmp_path / ("status_%d.npy" % r)), want.status[idx])
        assert np.array_equal(np.load(tmp_path / ("gm_%d.npy" % r)), want.group_metrics[idx])
    assert seen.all()
    whole = avk_dist.result_checksum(batch, want)
    assert int(np.load(tmp_path / "chk_0.npy")[0]) == whole == int(np.load(tmp_path / "chk_1.npy")[0])
    broken = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    broken.var_observed[3] ^= 1
    assert avk_dist.result_checksum(batch, broken) != whole  # one flipped decision changes the checksum
    sizes = [len(np.load(tmp_path / ("idx_%d.npy" % r))) for r in range(world)]
    assert abs(sizes[0] - sizes[1]) < 0.1 * batch.n_regions  # the hash balances the shards


def test_shards_have_the_packed_form():
    """bench.py --scaling strong hands every rank's shard over in the packed form (avk_packed_batch): gather_calls leaves a shard's calls and alleles back to back in
    region order, which is all that form asks for"""
    from aardvark_amd import CompactBatch, PackedBatch, synth
    from aardvark_amd import dist as avk_dist
    contigs, batch = synth.config_genome(scale=0.004)
    seen = 0
    for rank in range(3):
        shard = avk_dist.gather_calls(avk_dist.shard_batch(batch, rank, 3))
        pk = PackedBatch.from_compact(CompactBatch.from_region_batch(shard))
        assert pk.n_regions == shard.n_regions and pk.n_variants == shard.n_variants
        seen += pk.n_regions
    assert seen == batch.n_regions
