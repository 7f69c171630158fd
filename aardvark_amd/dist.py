"""Multi-GPU plumbing: independent regions are sharded by a hash of region_id, every rank solves
its shard with no data-path collective, and the fixed-size tally block (13 groups x 22 counters
+ solved/error blocks, reference src/writers/summary.rs:146-163) is summed with one all-reduce
(RCCL over xGMI on the GPU box, gloo in the CPU tests).  Integer sums: order independent, so the
reduced tally is bit-identical to a single-process run."""
import numpy as np

from ._abi import RegionBatch


def region_hash(region_id):
    """splitmix64 finaliser: spreads sequential region ids (genome order) over the ranks"""
    x = np.asarray(region_id, dtype=np.uint64).copy()
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def shard_indices(region_id, rank, world):
    return np.nonzero(region_hash(region_id) % np.uint64(world) == np.uint64(rank))[0]


def take_regions(batch, idx):
    """sub-batch of the selected regions (variant arrays are shared, not copied)"""
    return RegionBatch(batch.region_id[idx], batch.contig_idx[idx], batch.start[idx], batch.end[idx], batch.t_off[idx], batch.t_cnt[idx],
                       batch.q_off[idx], batch.q_cnt[idx], batch.var_pos, batch.var_type, batch.var_zyg, batch.var_raw_space,
                       batch.a0_off, batch.a0_len, batch.a1_off, batch.a1_len, batch.allele_bytes)


def shard_batch(batch, rank, world):
    """the regions of `batch` this rank owns: hash(region_id) % world == rank (reference loop src/main.rs:251-268 maps over
    independent regions).  bench.py --scaling strong and the tools use this one function."""
    if world <= 1:
        return batch
    return take_regions(batch, shard_indices(batch.region_id, rank, world))


def allreduce_tally(tally):
    """in-place SUM all-reduce of an int64 tally tensor over the default process group"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM)
    return tally
