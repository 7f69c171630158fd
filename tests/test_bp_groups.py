"""Compact per-region BASEPAIR groups (avk_result_batch::bp_off / bp_groups: 16 bytes per group instead of the 1144-byte block): the kernels' groups, expanded with
the per-call outputs by the library's host function avk_group_metrics_from_compact, must give the oracle's full GroupTypeMetrics block for every solved region."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import CompareConfig, synth
from aardvark_amd.api import group_metrics_from_compact


def check(batch, got, want):
    assert np.array_equal(got.status, want.status)
    full = group_metrics_from_compact(batch, got)
    ok = want.status == 0
    assert np.array_equal(full[ok], want.group_metrics[ok])
    # regions that fail validation own no group; the others 1 + their call types
    n_groups = np.diff(got.bp_off.astype(np.int64))
    for r in range(batch.n_regions):
        types = set(int(batch.var_type[v]) for off, cnt in ((batch.t_off[r], batch.t_cnt[r]), (batch.q_off[r], batch.q_cnt[r])) for v in range(int(off), int(off) + int(cnt)))
        assert n_groups[r] in (0, 1 + len(types)), r
        if want.status[r] == 0:
            assert n_groups[r] == 1 + len(types)


def check_packed(batch, got, want):
    """the packed form of the groups (avk_result_batch::bp_packed): one word per region, the others' groups spilled; same full blocks"""
    assert np.array_equal(got.status, want.status)
    full = group_metrics_from_compact(batch, got)
    ok = want.status == 0
    assert np.array_equal(full[ok], want.group_metrics[ok])
    spilled = (got.bp_packed[:batch.n_regions] & np.uint32(0x80000000)) != 0
    assert not spilled[~ok].any() and (got.bp_packed[:batch.n_regions][~ok] == 0).all()
    # a region with calls of one type and small counters is one word
    for r in np.nonzero(ok)[0]:
        types = set(int(batch.var_type[v]) for off, cnt in ((batch.t_off[r], batch.t_cnt[r]), (batch.q_off[r], batch.q_cnt[r])) for v in range(int(off), int(off) + int(cnt)))
        bp = want.group_metrics[r][0][18:22] if False else None
        if len(types) == 1 and int(want.group_metrics[r, 0].max()) < 128:
            assert not spilled[r], r
    assert int(got.bp_spilled[0]) == sum(1 + len(set(int(batch.var_type[v]) for off, cnt in ((batch.t_off[r], batch.t_cnt[r]), (batch.q_off[r], batch.q_cnt[r]))
                                                         for v in range(int(off), int(off) + int(cnt)))) for r in np.nonzero(spilled)[0])


def cases(light=False, emulator=False):
    """light: without the long alleles (counters over 127: spilled groups) and with fewer fuzz regions; emulator: alleles of up to 400 bases stand in for the
    10,000-base ones (a minute per run there)"""
    yield scenarios.golden()
    yield scenarios.fuzz_regions(71, 100 if light else 240, max_vars=6, max_len=10)
    yield scenarios.fuzz_regions(72, 160, max_vars=3, repeat_unit=b"CA")
    if not light:
        if emulator:
            yield scenarios.fuzz_regions(73, 16, max_vars=2, max_len=400, span=(300, 900))  # joint counters up to 1,443
        else:
            c = scenarios.long_allele_regions()
            yield c[0], c[1]
    yield scenarios.invalid_regions()
    contig, batch = synth.config_indel_mix_v2(n_truth=700, contig_len=400_000)
    yield [contig], batch


@pytest.mark.parametrize("lane_kernel", [True, False])
def test_compact_groups_rebuild_the_full_block_kernel_logic(oracle, lane_kernel):
    lib = emu_lib.load()
    lib.emu_set_device_pack.argtypes = [__import__("ctypes").c_int]
    for devpack in (0, 2):
        lib.emu_set_device_pack(devpack)
        try:
            for contigs, batch in cases(light=not (devpack == 2 and lane_kernel), emulator=True):  # the full set once, through the device packer and the lane code
                want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
                got = emu_lib.compare_batch(batch, contigs, lane_kernel=lane_kernel, group_metrics=False, bp_groups=True, threads=8)
                check(batch, got, want)
                if devpack == 2:  # the packed form is made by the device packer's result kernel
                    check_packed(batch, emu_lib.compare_batch(batch, contigs, lane_kernel=lane_kernel, group_metrics=False, bp_groups="packed", threads=8), want)
        finally:
            lib.emu_set_device_pack(0)


@pytest.mark.gpu
@pytest.mark.parametrize("device_pack", [1, 0])
def test_compact_groups_on_the_gpu(oracle, device_pack):
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("device_pack", device_pack)
        ctx.set_option("lane_min_regions", 0)
        ctx.set_option("lane_min_batch", 0)
        for lane in (1, 0):
            ctx.set_option("lane_kernel", lane)
            for contigs, batch in cases():
                want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8)
                ctx.upload_reference(contigs)
                got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, bp_groups=True)
                check(batch, got, want)
                if device_pack:
                    check_packed(batch, ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, bp_groups="packed"), want)
        # a contig at the benchmark's density: the compact groups cost 16 B x (1 + types) per region
        contig, batch = synth.config_indel_mix_v2(n_truth=60_000, contig_len=24_000_000)
        ctx.set_option("lane_kernel", 1)
        ctx.upload_reference([contig])
        want = oracle_lib.compare_batch(oracle, batch, [contig], threads=8)
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, bp_groups=True)
        check(batch, got, want)
        assert got.bp_off[-1] * 16 < 0.04 * want.group_metrics.nbytes
        if device_pack:
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, bp_groups="packed", packed=True)
            check_packed(batch, got, want)
            assert 4 * batch.n_regions + 16 * int(got.bp_spilled[0]) < 8 * batch.n_regions  # under 8 bytes per region
    finally:
        ctx.close()
