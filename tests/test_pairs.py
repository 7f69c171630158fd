"""Regions with the same SNV on both sides are looked up in a table that the solver itself fills (aardvark_amd/csrc/avk_pairs.inl): every
(truth zygosity, query zygosity) pair, in windows that try to make the bases matter — homopolymers and tandem repeats around the call, calls at
the first and last base of the window, lower-case and N bases in the window (under the call: the region is handed to the wave-per-region
code), an ALT base equal to the reference base (the caller's REF disagrees with the genome: handed over as well), branch quotas from 1 up —
against the oracle, bit for bit, with the lookup on and off."""
import ctypes as C

import numpy as np
import pytest

import emu_lib
import oracle_lib
from aardvark_amd import CompareConfig, RegionBatch, synth

ZY = ["UnphasedHeterozygous", "PhasedHet01", "PhasedHet10", "HomozygousAlternate"]


def pair_regions(seed, reps=3):
    rng = np.random.default_rng(seed)
    parts, regions, at = [], [], 0

    def add(window, pos, alt, zt, zq, ref_base=None):
        nonlocal at
        w = bytearray(window)
        ref = bytes([w[pos]]).upper() if ref_base is None else ref_base
        parts.append(bytes(w))
        v = lambda z: (at + pos, ref, alt, "Snv", z)
        regions.append({"start": at, "end": at + len(w), "truth": [v(zt)], "query": [v(zq)]})
        at += len(w)

    def other(b):
        return bytes([next(x for x in b"ACGT" if x != b)])

    for zt in ZY:
        for zq in ZY:
            for _ in range(reps):
                L = int(rng.integers(8, 180))
                w = bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))
                pos = int(rng.integers(0, L))
                add(w, pos, other(w[pos]), zt, zq)
            add(b"A" * 101, 50, b"C", zt, zq)                      # homopolymer
            add(b"CA" * 60, 61, b"G", zt, zq)                      # tandem repeat
            add(b"ACGTTGCA" * 10, 0, b"C", zt, zq)                 # first base of the window
            add(b"ACGTTGCA" * 10, 79, b"C", zt, zq)                # last base
            add(b"ACGTNNNNacgtACGTACGTACGTTTGA", 20, b"G", zt, zq) # other symbols in the window, not under the call
            add(b"ACGTACGTacgtACGTACGT", 9, b"G", zt, zq)          # a lower-case base under the call
            add(b"ACGTACGTANGTACGTACGT", 9, b"G", zt, zq, ref_base=b"C")  # N under the call
            add(b"ACGTACGTACGTACGTACGT", 9, b"C", zt, zq, ref_base=b"G")  # ALT equals the genome's base: the caller's REF is something else
    contig = b"".join(parts)
    return [contig], RegionBatch.from_regions(regions)


def lib_pairs(on):
    lib = emu_lib.load()
    lib.emu_set_lane_pairs.argtypes = [C.c_int]
    lib.emu_last_pair_regions.restype = C.c_uint64
    lib.emu_set_lane_pairs(1 if on else 0)
    return lib


@pytest.mark.parametrize("devpack", [0, 2])
@pytest.mark.parametrize("quota", [50, 1, 2])
def test_every_zygosity_pair_kernel_logic(oracle, devpack, quota):
    contigs, batch = pair_regions(5)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4, max_branch_factor=quota)
    lib = lib_pairs(True)
    lib.emu_set_device_pack.argtypes = [C.c_int]
    lib.emu_set_device_pack(devpack)
    try:
        on = emu_lib.compare_batch(batch, contigs, threads=8, max_branch_factor=quota, bp_groups=True)
        n_pairs = int(lib.emu_last_pair_regions())
        lib_pairs(False)
        off = emu_lib.compare_batch(batch, contigs, threads=8, max_branch_factor=quota, bp_groups=True)
        assert int(lib.emu_last_pair_regions()) == 0
    finally:
        lib_pairs(True)
        lib.emu_set_device_pack(0)
    assert on.diff(want) == []
    assert off.diff(want) == []
    assert n_pairs == batch.n_regions  # every region is a candidate; the ones the table cannot answer are handed over by the lookup itself
    assert np.array_equal(on.tally, off.tally) and np.array_equal(on.tally, want.tally)
    assert np.array_equal(on.bp_off, off.bp_off) and np.array_equal(on.bp_groups[:on.bp_off[-1]], off.bp_groups[:off.bp_off[-1]])
    # the statuses are not all the same thing: zygosity pairs differ in their optima and their flips
    assert len({(int(a), int(b), int(c)) for a, b, c in zip(want.ed_h1, want.ed_h2, want.n_optima)}) >= 3


def test_benchmark_mix_uses_the_lookup(oracle):
    contig, batch = synth.config_indel_mix_v2(n_truth=4000, contig_len=2_000_000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=4)
    lib = lib_pairs(True)
    got = emu_lib.compare_batch(batch, [contig], threads=8)
    assert got.diff(want) == []
    assert int(lib.emu_last_pair_regions()) > 0.5 * batch.n_regions
    assert np.array_equal(got.tally, want.tally)


@pytest.mark.gpu
@pytest.mark.parametrize("device_pack", [1, 0])
def test_every_zygosity_pair_on_the_gpu(oracle, device_pack):
    import aardvark_amd
    contigs, batch = pair_regions(7, reps=40)
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("device_pack", device_pack)
        ctx.set_option("lane_min_regions", 0)
        ctx.set_option("lane_min_batch", 0)
        ctx.upload_reference(contigs)
        for quota in (50, 1, 3):
            want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8, max_branch_factor=quota)
            res = {}
            for pairs in (1, 0):
                ctx.set_option("lane_pairs", pairs)
                res[pairs] = ctx.solve_compare_regions(batch, CompareConfig(max_branch_factor=quota, enable_sequences=False), bp_groups=True)
                assert res[pairs].diff(want) == []
                assert np.array_equal(res[pairs].tally, want.tally)
            assert np.array_equal(res[1].bp_off, res[0].bp_off) and np.array_equal(res[1].bp_groups[:res[1].bp_off[-1]], res[0].bp_groups[:res[0].bp_off[-1]])
            assert ctx.last_lane_solved() >= 0.9 * batch.n_regions
        # the full metric blocks come from the table too
        ctx.set_option("lane_pairs", 1)
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=True)
        want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8)
        assert got.diff(want) == [] and np.array_equal(got.group_metrics, want.group_metrics)
    finally:
        ctx.close()


def ref_mismatch_regions(seed, n):
    """calls whose REF allele is not what the contig has at their position (first base changed; half of them with the ALT's first base changed along with it, the
    way a normalised indel shares its anchor): the haplotypes are spliced from the WINDOW (generate_allele_sequence, waffle_solver.rs:726-778), so no distance may
    be taken from the alleles alone"""
    import scenarios
    rng = np.random.default_rng(seed)
    contig_len = 6000
    contig = np.frombuffer(bytes(rng.choice(list(b"ACGT"), size=contig_len).astype(np.uint8)), dtype=np.uint8).copy()

    def twist(v):
        pos, ref, alt, vt, z = v
        u = rng.random()
        if u < 0.4:
            return v
        nb = bytes([next(x for x in b"ACGT" if x != ref[0])])
        if u < 0.7 or len(alt) == 0:
            return (pos, nb + ref[1:], alt, vt, z)
        if len(ref) == 1 and len(alt) == 1 and u < 0.85:
            return (pos, nb, bytes([contig[pos]]), vt, z)  # an SNV whose ALT is the genome's base
        return (pos, nb + ref[1:], nb + alt[1:], vt, z)

    regions = []
    for _ in range(n):
        L = int(rng.integers(12, 150))
        start = int(rng.integers(0, contig_len - L))
        nt = int(rng.integers(1, 3))
        truth = sorted([twist(scenarios.random_variant(rng, contig, start, start + L, 6)) for _ in range(nt)], key=lambda v: v[0])
        query = []
        for v in truth:
            u = rng.random()
            if u < 0.2:
                continue
            query.append(v if u < 0.7 else (v[0], v[1], v[2], v[3], ZY[int(rng.integers(0, 4))]))
        if rng.random() < 0.3:
            query.append(twist(scenarios.random_variant(rng, contig, start, start + L, 6)))
        query.sort(key=lambda v: v[0])
        regions.append({"start": start, "end": start + L, "truth": truth, "query": query})
    return [bytes(contig)], RegionBatch.from_regions(regions)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_ref_alleles_that_disagree_with_the_genome_kernel_logic(oracle, seed):
    contigs, batch = ref_mismatch_regions(seed, 500)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    lib_pairs(True)
    lane = emu_lib.compare_batch(batch, contigs, threads=8, lane_kernel=True)
    wave = emu_lib.compare_batch(batch, contigs, threads=8, lane_kernel=False)
    assert lane.diff(want) == []
    assert wave.diff(want) == []
    assert lane.lane_solved > 0.5 * batch.n_regions


@pytest.mark.gpu
def test_ref_alleles_that_disagree_with_the_genome_on_the_gpu(oracle):
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("lane_min_regions", 0)
        ctx.set_option("lane_min_batch", 0)
        for seed in (11, 12, 13, 14):
            contigs, batch = ref_mismatch_regions(seed, 3000)
            want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8)
            ctx.upload_reference(contigs)
            for lane in (1, 0):
                ctx.set_option("lane_kernel", lane)
                got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False))
                assert got.diff(want) == [], (seed, lane)
    finally:
        ctx.close()
