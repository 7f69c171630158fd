"""The device-side batch packer (aardvark_amd/csrc/avk_devpack.inl: validation, Variant::alt_ed, lane classes and cost keys, work plan, work order, fast
records, region records and blobs — all as kernels on the caller's arrays) against the host-side packer it replaces (avk_pack.h), record by record, and
end to end against the oracle.  The device functions run here through the kernel-logic emulator (tests/emu: the same source, the workgroup plumbing as
plain loops); the GPU runs of the same code are in test_gpu_devpack.py."""
import ctypes as C

import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import RegionBatch, synth
from aardvark_amd._abi import AvkRegionBatch
from oracle_lib import u8p, u64p


def _lib():
    lib = emu_lib.load()
    lib.emu_devpack_compare.argtypes = [C.POINTER(AvkRegionBatch), u64p, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    lib.emu_devpack_alt_ed.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.POINTER(C.c_int)]
    lib.emu_devpack_alt_ed.restype = C.c_uint32
    lib.emu_set_device_pack.argtypes = [C.c_int]
    return lib


def same_as_host_packer(batch, contig_lens, pairs=False, lane_min_regions=0, lane_min_batch=0, lane_max_est=15, solo_min_variants=5):
    lib = _lib()
    lens = np.asarray(contig_lens, np.uint64)
    msg = C.create_string_buffer(2048)
    cb = batch.c_struct()
    rc = lib.emu_devpack_compare(C.byref(cb), lens.ctypes.data_as(u64p), len(lens), 1 if pairs else 0, lane_min_regions, lane_min_batch, lane_max_est, solo_min_variants, msg, 2048)
    assert rc == 0, msg.value.decode()


def lens_of(contigs):
    return [len(c) for c in contigs]


def test_alt_ed_equals_host_edit_distance():
    """dp_variant's alt_ed (prefix / suffix strip + the 64-bit bit-vector recurrence) == avk_edit_distance (the host packer's, == oracle wfa_ed)"""
    import aardvark_amd
    lib = _lib()
    host = aardvark_amd.load_library()
    rng = np.random.default_rng(7)
    pend = C.c_int(0)
    n_checked = 0
    for it in range(4000):
        kind = it % 5
        if kind == 0:  # unrelated strings, lengths 1..70
            a = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 70))).astype(np.uint8))
            b = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 70))).astype(np.uint8))
        elif kind == 1:  # a mutated copy (few edits), up to 300 symbols
            a = bytearray(rng.choice(list(b"ACGT"), size=int(rng.integers(2, 300))).astype(np.uint8))
            b = bytearray(a)
            for _ in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, len(b)))
                op = int(rng.integers(0, 3))
                if op == 0:
                    b[p] = b"ACGT"[int(rng.integers(0, 4))]
                elif op == 1 and len(b) > 1:
                    del b[p]
                else:
                    b.insert(p, b"ACGT"[int(rng.integers(0, 4))])
            a, b = bytes(a), bytes(b)
        elif kind == 2:  # two-letter alphabet: many equally good alignments
            a = bytes(rng.choice(list(b"AC"), size=int(rng.integers(1, 64))).astype(np.uint8))
            b = bytes(rng.choice(list(b"AC"), size=int(rng.integers(1, 200))).astype(np.uint8))
        elif kind == 3:  # exactly 64 / 65 symbols on the short side (the width of the bit vector)
            a = bytes(rng.choice(list(b"ACGTN"), size=int(rng.choice([63, 64, 65]))).astype(np.uint8))
            b = bytes(rng.choice(list(b"ACGTN"), size=int(rng.integers(60, 120))).astype(np.uint8))
        else:  # normalised indels
            a = bytes(rng.choice(list(b"ACGT"), size=1).astype(np.uint8))
            b = a + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 40))).astype(np.uint8))
            if it % 2:
                a, b = b, a
        a0 = np.frombuffer(a, np.uint8)
        a1 = np.frombuffer(b, np.uint8)
        got = lib.emu_devpack_alt_ed(a0.ctypes.data_as(u8p), len(a), a1.ctypes.data_as(u8p), len(b), C.byref(pend))
        if pend.value:
            continue  # both stripped alleles longer than 64: left to the host by design
        want = host.avk_edit_distance(a, len(a), b, len(b))
        assert got == want, (a, b, got, want)
        n_checked += 1
    assert n_checked > 3500


def test_reference_known_answer_regions():
    contigs, batch = scenarios.golden()
    same_as_host_packer(batch, lens_of(contigs))
    same_as_host_packer(batch, lens_of(contigs), lane_min_regions=0xFFFFFFFF)
    same_as_host_packer(batch, lens_of(contigs), pairs=True)


@pytest.mark.parametrize("seed,kw", [(201, {}), (202, {"repeat_unit": b"CA", "max_vars": 3}), (203, {"max_vars": 9, "max_len": 12}), (204, {"max_len": 40, "span": (20, 250)}),
                                     (205, {"alphabet": b"ACGTN", "max_vars": 3})])
def test_fuzz_regions_pack_identically(seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 1500, **kw)
    same_as_host_packer(batch, lens_of(contigs))
    same_as_host_packer(batch, lens_of(contigs), lane_max_est=3, solo_min_variants=3)
    same_as_host_packer(batch, lens_of(contigs), pairs=True)


def test_benchmark_mix_packs_identically():
    """the features of the benchmark genome; with the production thresholds for the lane classes (some classes too small for a launch) and without"""
    contig, batch = synth.config_indel_mix_v2(n_truth=30_000, contig_len=12_000_000)
    same_as_host_packer(batch, [len(contig)])
    same_as_host_packer(batch, [len(contig)], lane_min_regions=512, lane_min_batch=1000)
    same_as_host_packer(batch, [len(contig)], lane_min_regions=8192, lane_min_batch=65536)


@pytest.mark.parametrize("below", [1 << 20, 10])
def test_few_regions_outside_the_lanes_are_planned_for_the_wide_kernel(oracle, below):
    """context option class_c_below: a batch with lane launches and at most that many regions outside them plans those the wide kernel can take as class C; both
    packers make the same plan under it, more regions reach the wide kernel than without it, and the results do not change"""
    lib = _lib()
    lib.emu_set_class_c_below.argtypes = [C.c_uint64]
    contig, batch = synth.config_indel_mix_v2(n_truth=6000, contig_len=2_400_000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=4)
    off = emu_lib.compare_batch(batch, [contig], n_waves=16)
    lib.emu_set_class_c_below(below)
    try:
        same_as_host_packer(batch, [len(contig)])
        same_as_host_packer(batch, [len(contig)], lane_min_regions=256, lane_min_batch=1000)
        same_as_host_packer(batch, [len(contig)], lane_min_regions=1 << 20)  # no lane launches: the rule does not apply
        for mode in (0, 2):
            lib.emu_set_device_pack(mode)
            got = emu_lib.compare_batch(batch, [contig], n_waves=16)
            assert got.diff(want) == []
            outside = batch.n_regions - got.lane_solved
            if below >= outside:
                assert got.wide_solved > off.wide_solved and got.wide_solved > 0.5 * outside
            else:
                assert got.wide_solved == off.wide_solved
    finally:
        lib.emu_set_device_pack(0)
        lib.emu_set_class_c_below(0)


def test_invalid_long_and_odd_inputs_pack_identically():
    for sc in (scenarios.invalid_regions(), scenarios.long_allele_regions(), scenarios.non_acgt_regions(), scenarios.autofail_regions(), scenarios.quota_regions(3),
               scenarios.max_allele_regions(), scenarios.optimizer_golden_regions()):
        contigs, batch = sc[0], sc[1]
        same_as_host_packer(batch, lens_of(contigs))
        same_as_host_packer(batch, lens_of(contigs), pairs=True)


def large_regions_batch():
    rng = np.random.default_rng(11)
    contig = bytes(rng.choice(list(b"ACGT"), size=60_000).astype(np.uint8))
    regions = []
    for k in range(6):
        start, end = 1000 + 9000 * k, 1000 + 9000 * k + 8000
        def calls(n):
            pos = np.sort(rng.choice(np.arange(start + 5, end - 40), size=n, replace=k % 2 == 0))
            out = []
            for p in pos:
                p = int(p)
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    out.append((p, contig[p:p + 1], bytes([b"ACGT"[(b"ACGT".index(contig[p]) + 1) % 4]]), "Snv", scenarios.ZY[int(rng.integers(0, 4))]))
                elif kind == 1:
                    out.append((p, contig[p:p + 1], contig[p:p + 1] + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 30))).astype(np.uint8)), "Insertion", scenarios.ZY[int(rng.integers(0, 4))]))
                else:
                    d = int(rng.integers(2, 20))
                    out.append((p, contig[p:p + d], contig[p:p + 1], "Deletion", scenarios.ZY[int(rng.integers(0, 4))]))
            return out
        regions.append({"start": start, "end": end, "truth": calls(int(rng.integers(49, 300))), "query": calls(int(rng.integers(1, 300)))})
    return contig, RegionBatch.from_regions(regions)


def test_large_regions_go_through_the_wave_writer():
    """more than 48 calls in a region: the blob is written by a whole wave (prefix sums over lanes, ranks by binary search)"""
    contig, batch = large_regions_batch()
    same_as_host_packer(batch, [len(contig)])


def test_batch_level_errors_match():
    contigs, batch = scenarios.fuzz_regions(31, 50)
    bad = RegionBatch(batch.region_id, batch.contig_idx, batch.start, batch.end, batch.t_off, batch.t_cnt.copy(), batch.q_off, batch.q_cnt, batch.var_pos, batch.var_type,
                      batch.var_zyg, batch.var_raw_space, batch.a0_off, batch.a0_len, batch.a1_off, batch.a1_len, batch.allele_bytes)
    bad.t_cnt[7] = 0x80000000
    same_as_host_packer(bad, lens_of(contigs))
    bad2 = RegionBatch(batch.region_id, batch.contig_idx, batch.start, batch.end, batch.t_off, batch.t_cnt, batch.q_off, batch.q_cnt, batch.var_pos, batch.var_type,
                       batch.var_zyg, batch.var_raw_space, batch.a0_off, batch.a0_len, batch.a1_off.copy(), batch.a1_len, batch.allele_bytes)
    used = int(batch.t_off[3]) if batch.t_cnt[3] else int(batch.q_off[3])
    bad2.a1_off[used] = batch.allele_bytes.size + 5
    same_as_host_packer(bad2, lens_of(contigs))


@pytest.mark.parametrize("lane_kernel,mode", [(True, 1), (True, 2), (False, 2)])
def test_end_to_end_through_the_emulator_equals_oracle(oracle, lane_kernel, mode):
    """device-packed batch -> solver kernels -> dp_unpack: every output array equals the oracle's.  mode 2 = as upload_device_packed does it: region
    records and blobs of the lane classes' regions are only written for the regions the lanes hand back, right before the launch that solves them"""
    lib = _lib()
    lib.emu_set_device_pack(mode)
    try:
        contigs, batch = scenarios.golden()
        got = emu_lib.compare_batch(batch, contigs, lane_kernel=lane_kernel, n_waves=2)
        assert got.diff(oracle_lib.compare_batch(oracle, batch, contigs)) == []
        contig, batch = synth.config_indel_mix_v2(n_truth=3000, contig_len=1_500_000)
        got = emu_lib.compare_batch(batch, [contig], lane_kernel=lane_kernel, n_waves=16)
        assert got.diff(oracle_lib.compare_batch(oracle, batch, [contig], threads=4)) == []
        if lane_kernel:
            assert got.lane_solved > 0.9 * batch.n_regions
        for contigs, batch in (scenarios.fuzz_regions(41, 300, max_vars=6), scenarios.invalid_regions(), scenarios.non_acgt_regions()):
            got = emu_lib.compare_batch(batch, contigs, lane_kernel=lane_kernel)
            assert got.diff(oracle_lib.compare_batch(oracle, batch, contigs, threads=4)) == []
        contigs, batch = scenarios.fuzz_regions(43, 300, max_vars=3, related=0.9)
        st, ex = emu_lib.optimize_pairs(batch, contigs)
        lib.emu_set_device_pack(0)
        st0, ex0 = emu_lib.optimize_pairs(batch, contigs)
        assert np.array_equal(st, st0) and np.array_equal(ex, ex0)
    finally:
        lib.emu_set_device_pack(0)


def test_long_unrelated_alleles_take_the_host_edit_distance():
    """both alleles longer than 64 symbols after the common prefix and suffix are gone: alt_ed comes from the host and the region passes run again"""
    rng = np.random.default_rng(5)
    contig = bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8))
    regions = []
    for k in range(5):
        start = 200 + 900 * k
        p = start + 50
        ref = contig[p:p + 150 + 10 * k]
        alt = bytes(rng.choice(list(b"ACGT"), size=120 + 7 * k).astype(np.uint8))
        v = (p, ref, alt, "Indel", "HomozygousAlternate")
        regions.append({"start": start, "end": start + 600, "truth": [v, (p + 300, contig[p + 300:p + 301], b"T" if contig[p + 300] != ord("T") else b"G", "Snv", "UnphasedHeterozygous")],
                        "query": [v] if k % 2 else []})
    batch = RegionBatch.from_regions(regions)
    lib = _lib()
    a0 = np.frombuffer(regions[0]["truth"][0][1], np.uint8)
    a1 = np.frombuffer(regions[0]["truth"][0][2], np.uint8)
    pend = C.c_int(0)
    lib.emu_devpack_alt_ed(a0.ctypes.data_as(u8p), len(a0), a1.ctypes.data_as(u8p), len(a1), C.byref(pend))
    assert pend.value == 1
    same_as_host_packer(batch, [len(contig)])


def test_compact_form_round_trip():
    """CompactBatch.from_region_batch / widen (the numpy statement of the library's dp_widen kernel): the wide arrays come back as they were"""
    from aardvark_amd import CompactBatch
    contig, batch = synth.config_indel_mix_v2(n_truth=5000, contig_len=2_000_000)
    cb = CompactBatch.from_region_batch(batch)
    w = cb.widen()
    for f in ("contig_idx", "start", "end", "t_off", "t_cnt", "q_off", "q_cnt", "var_pos", "var_type", "var_zyg", "var_raw_space", "a0_off", "a0_len", "a1_off", "a1_len", "allele_bytes"):
        assert np.array_equal(getattr(w, f), getattr(batch, f)), f
    assert cb.nbytes() < 0.55 * 90 * batch.n_variants + 60 * batch.n_regions
    contigs, bad = scenarios.invalid_regions()
    with pytest.raises(ValueError):
        CompactBatch.from_region_batch(RegionBatch(bad.region_id, bad.contig_idx, bad.start, bad.end, bad.t_off, bad.t_cnt, bad.q_off + 1, bad.q_cnt, bad.var_pos, bad.var_type, bad.var_zyg,
                                                   bad.var_raw_space, bad.a0_off, bad.a0_len, bad.a1_off, bad.a1_len, bad.allele_bytes))


def test_packed_form_round_trip_and_constraints():
    """PackedBatch.from_compact: the offsets the packed form leaves out are exactly the running sums of its counts and lengths; batches that break a constraint
    (calls out of region order, an allele of 256 bases, a window of 65,536) have no packed form"""
    from aardvark_amd import CompactBatch, PackedBatch, synth
    contig, batch = synth.config_indel_mix_v2(n_truth=3000, contig_len=1_500_000)
    cb = CompactBatch.from_region_batch(batch)
    pk = PackedBatch.from_compact(cb)
    assert pk.n_regions == cb.n_regions and pk.n_variants == cb.n_variants and pk.nbytes() < 0.5 * cb.nbytes()
    cnt = pk.t_cnt.astype(np.int64) + pk.q_cnt
    assert np.array_equal(np.concatenate([[0], np.cumsum(cnt)])[:-1], cb.v_off)
    assert np.array_equal(np.repeat(pk.start.astype(np.int64), cnt) + pk.var_rel_pos, cb.var_pos)
    assert np.array_equal(np.concatenate([[0], np.cumsum(pk.a0_len.astype(np.int64) + pk.a1_len)])[:-1], cb.a_off)
    import copy
    for breaker in ("v_off", "a0_len", "len"):
        c2 = copy.deepcopy(cb)
        if breaker == "v_off":
            c2.v_off = c2.v_off.copy()
            c2.v_off[3], c2.v_off[4] = c2.v_off[4], c2.v_off[3]
        elif breaker == "a0_len":
            c2.a0_len = c2.a0_len.copy()
            c2.a0_len[7] = 256
        else:
            c2.len = c2.len.copy()
            c2.len[2] = 65536
        with pytest.raises(ValueError):
            PackedBatch.from_compact(c2)


def test_packed_multi_form_round_trip_and_constraints():
    """PackedMultiBatch.from_multi / widen: the offsets the packed form leaves out are the running sums of its counts and lengths, the wide batch comes back field
    by field; batches that break a constraint have no packed form"""
    import copy
    from aardvark_amd import synth
    from aardvark_amd.merge import MultiBatch, PackedMultiBatch
    contigs, mb = synth.config_genome_merge(scale=0.004, k=3, threads=4)
    pm = PackedMultiBatch.from_multi(mb)
    assert pm.n_regions == mb.n_regions and pm.n_variants == mb.n_variants and pm.n_inputs == 3
    assert pm.nbytes() < 0.25 * sum(getattr(mb, f).nbytes for f in MultiBatch.FIELDS)
    wide = pm.widen()
    for f in MultiBatch.FIELDS:
        if f != "region_id":
            assert np.array_equal(getattr(wide, f)[:getattr(mb, f).size], getattr(mb, f)), f
    kept = PackedMultiBatch.from_multi(mb, keep_raw_space=True)
    assert pm.var_raw_space is None and np.array_equal(kept.var_raw_space, mb.var_raw_space)
    for breaker in ("in_off", "a0_len", "end", "in_cnt"):
        m2 = copy.deepcopy(mb)
        if breaker == "in_off":
            m2.in_off = m2.in_off.copy()
            m2.in_off[3], m2.in_off[4] = m2.in_off[4] + 1, m2.in_off[3]
        elif breaker == "a0_len":
            m2.a0_len = m2.a0_len.copy()
            m2.a0_len[7] = 256
        elif breaker == "end":
            m2.end = m2.end.copy()
            m2.end[2] = m2.start[2] + 65536
        else:
            m2.in_cnt = m2.in_cnt.copy()
            m2.in_cnt[5] += 1
        with pytest.raises(ValueError):
            PackedMultiBatch.from_multi(m2)


# ---- round 6: a batch that came in the packed form is packed FROM the packed arrays (DpIn::pk_*: every offset implied by order, positions relative to the region's
# start): no wide copy of the caller's arrays is made in HBM.  Same records, same plan, same results.

@pytest.fixture
def packed_source():
    lib = _lib()
    lib.emu_set_packed_source.argtypes = [C.c_int]
    lib.emu_set_packed_source(1)
    yield lib
    lib.emu_set_packed_source(0)


def test_packed_source_packs_identically(packed_source):
    lib = packed_source
    contigs, batch = scenarios.golden()
    same_as_host_packer(batch, lens_of(contigs))
    assert lib.emu_last_packed_source() == 1
    same_as_host_packer(batch, lens_of(contigs), lane_min_regions=0xFFFFFFFF)
    for seed, kw in [(201, {}), (202, {"repeat_unit": b"CA", "max_vars": 3}), (203, {"max_vars": 9, "max_len": 12}), (204, {"max_len": 40, "span": (20, 250)}),
                     (205, {"alphabet": b"ACGTN", "max_vars": 3})]:
        contigs, batch = scenarios.fuzz_regions(seed, 800, **kw)
        lib.emu_set_packed_source(1)
        same_as_host_packer(batch, lens_of(contigs))
        assert lib.emu_last_packed_source() == 1, seed
        same_as_host_packer(batch, lens_of(contigs), lane_max_est=3, solo_min_variants=3)
    contig, batch = synth.config_indel_mix_v2(n_truth=20_000, contig_len=8_000_000)
    lib.emu_set_packed_source(1)
    same_as_host_packer(batch, [len(contig)])
    assert lib.emu_last_packed_source() == 1
    same_as_host_packer(batch, [len(contig)], lane_min_regions=512, lane_min_batch=1000)
    for sc in (scenarios.invalid_regions(), scenarios.long_allele_regions(), scenarios.non_acgt_regions(), scenarios.autofail_regions(), scenarios.quota_regions(3),
               scenarios.max_allele_regions(), scenarios.optimizer_golden_regions()):
        same_as_host_packer(sc[1], lens_of(sc[0]))  # (those that do not fit the form keep the wide path: same outcome either way)


def test_packed_source_large_regions_go_through_the_wave_writer(packed_source):
    contig, batch = large_regions_batch()
    assert int((batch.t_cnt.astype(np.int64) + batch.q_cnt).max()) > 255  # ... which does not fit the form's one-byte counts: the wide path stands
    same_as_host_packer(batch, [len(contig)])
    assert packed_source.emu_last_packed_source() == 0
    keep = np.nonzero((batch.t_cnt < 200) & (batch.q_cnt < 200) & (batch.t_cnt.astype(np.int64) + batch.q_cnt > 48))[0]
    assert keep.size >= 1
    from aardvark_amd import dist
    sub = dist.gather_calls(dist.take_regions(batch, keep))
    same_as_host_packer(sub, [len(contig)])
    assert packed_source.emu_last_packed_source() == 1


@pytest.mark.parametrize("mode", [1, 2])
def test_packed_source_end_to_end_through_the_emulator_equals_oracle(oracle, packed_source, mode):
    lib = packed_source
    lib.emu_set_device_pack(mode)
    try:
        contigs, batch = scenarios.golden()
        got = emu_lib.compare_batch(batch, contigs, n_waves=2)
        assert lib.emu_last_packed_source() == 1
        assert got.diff(oracle_lib.compare_batch(oracle, batch, contigs)) == []
        contig, batch = synth.config_indel_mix_v2(n_truth=3000, contig_len=1_500_000)
        got = emu_lib.compare_batch(batch, [contig], n_waves=16)
        assert lib.emu_last_packed_source() == 1
        assert got.diff(oracle_lib.compare_batch(oracle, batch, [contig], threads=4)) == []
        assert got.lane_solved > 0.9 * batch.n_regions
        for contigs, batch in (scenarios.fuzz_regions(41, 300, max_vars=6), scenarios.invalid_regions(), scenarios.non_acgt_regions()):
            got = emu_lib.compare_batch(batch, contigs)
            assert got.diff(oracle_lib.compare_batch(oracle, batch, contigs, threads=4)) == []
    finally:
        lib.emu_set_device_pack(0)
