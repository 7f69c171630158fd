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


def gather_calls(batch):
    """the same regions with call and allele arrays that hold only THEIR calls, region by region (truth calls, then query calls): what a rank copies to its GPU is
    its shard, not the job; also the layout the compact form (CompactBatch) asks for"""
    from .synth import _ragged
    n = batch.n_regions
    tc, qc = batch.t_cnt.astype(np.int64), batch.q_cnt.astype(np.int64)
    starts = np.stack([batch.t_off.astype(np.int64), batch.q_off.astype(np.int64)], axis=1).reshape(-1)
    lens = np.stack([tc, qc], axis=1).reshape(-1)
    src, _ = _ragged(starts, lens)
    new_t = np.cumsum(tc + qc) - (tc + qc)
    a0_len, a1_len = batch.a0_len[src].astype(np.int64), batch.a1_len[src].astype(np.int64)
    a0_off = np.cumsum(a0_len + a1_len) - (a0_len + a1_len)
    a1_off = a0_off + a0_len
    arena = np.zeros(int((a0_len + a1_len).sum()) + 1, np.uint8)
    d, _ = _ragged(a0_off, a0_len)
    sidx, _ = _ragged(batch.a0_off[src].astype(np.int64), a0_len)
    arena[d] = batch.allele_bytes[sidx]
    d, _ = _ragged(a1_off, a1_len)
    sidx, _ = _ragged(batch.a1_off[src].astype(np.int64), a1_len)
    arena[d] = batch.allele_bytes[sidx]
    return RegionBatch(batch.region_id, batch.contig_idx, batch.start, batch.end, new_t, batch.t_cnt, new_t + tc, batch.q_cnt, batch.var_pos[src], batch.var_type[src], batch.var_zyg[src],
                       batch.var_raw_space[src], a0_off, a0_len, a1_off, a1_len, arena[:max(int((a0_len + a1_len).sum()), 1)])


def shard_batch(batch, rank, world):
    """the regions of `batch` this rank owns: hash(region_id) % world == rank (reference loop src/main.rs:251-268 maps over
    independent regions).  bench.py --scaling strong and the tools use this one function."""
    if world <= 1:
        return batch
    return take_regions(batch, shard_indices(batch.region_id, rank, world))


def result_checksum(batch, res):
    """An order-independent 64-bit checksum of every per-region and per-variant output of `res` for the regions of `batch`: the sum, modulo
    2^64, of a hash per region (region_id, status, ed_h1, ed_h2, optima, types) and a hash per variant (region_id, index in the region,
    expected, observed, class, resolved zygosity).  The checksums of the shards of a job add up to the checksum of the whole job, so a
    sharded run can be compared with a single-process one by one integer (bench.py --scaling strong)."""
    with np.errstate(over="ignore"):
        rid = np.asarray(batch.region_id, np.uint64)
        w = (res.status.astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + res.ed_h1.astype(np.uint64) * np.uint64(1000003) +
             res.ed_h2.astype(np.uint64) * np.uint64(998244353) + res.n_optima.astype(np.uint64) * np.uint64(7919) + res.type_present.astype(np.uint64) * np.uint64(104729))
        total = region_hash(rid * np.uint64(3) + np.uint64(1) + w * np.uint64(0xD6E8FEB86659FD93)).sum(dtype=np.uint64)
        for side, off, cnt in ((np.uint64(1), batch.t_off, batch.t_cnt), (np.uint64(2), batch.q_off, batch.q_cnt)):
            cnt64 = np.asarray(cnt, np.int64)
            n = int(cnt64.sum())
            if n == 0:
                continue
            first = np.cumsum(cnt64) - cnt64
            reg = np.repeat(np.arange(batch.n_regions), cnt64)
            k = np.arange(n, dtype=np.int64) - np.repeat(first, cnt64)
            v = (np.repeat(np.asarray(off, np.int64), cnt64) + k)
            word = (res.var_expected[v].astype(np.uint64) | (res.var_observed[v].astype(np.uint64) << np.uint64(8)) |
                    (res.var_class[v].astype(np.uint64) << np.uint64(16)) | (res.var_zyg[v].astype(np.uint64) << np.uint64(24)))
            total = total + region_hash(rid[reg] * np.uint64(1315423911) + k.astype(np.uint64) * np.uint64(2654435761) + side * np.uint64(0x51ED27) +
                                        word * np.uint64(0x9E3779B97F4A7C15)).sum(dtype=np.uint64)
    return int(total)


def allreduce_tally(tally):
    """in-place SUM all-reduce of an int64 tally tensor over the default process group"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM)
    return tally


def allreduce_counts(counts):
    """in-place SUM all-reduce of an int64 tensor of sums over the default process group: a sharded merge's summary counters (aardvark_amd.merge.merge_counts; the
    reference's only state across merge regions, src/writers/merge_summary.rs:12-18)"""
    return allreduce_tally(counts)
