"""The device-side packer on the GPU (aardvark_amd/csrc/avk_devpack.inl behind avk_batch_upload / avk_compare_batch): the same batches through
device_pack = 1 (default) and device_pack = 0 (host packer, avk_pack.h) against the oracle; pinned and pageable caller arrays; the pair form; the
capacity retry and the sequence outputs of a device-packed batch (they fetch the packed records back).  tests/test_devpack.py runs the packer's
device functions on the CPU against the host packer record by record."""
import numpy as np
import pytest

import oracle_lib
import scenarios
from aardvark_amd import CompareConfig, RegionBatch, synth

pytestmark = pytest.mark.gpu
CPUS = 8


@pytest.fixture(scope="module")
def ctx_pair():
    import aardvark_amd
    dev, host = aardvark_amd.Context(0), aardvark_amd.Context(0)
    host.set_option("device_pack", 0)
    yield dev, host
    dev.close()
    host.close()


def check(ctx_pair, oracle, contigs, batch, sequences=False, group_metrics=True, **okw):
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, sequences=sequences, group_metrics=group_metrics, **okw)
    for c in ctx_pair:
        c.upload_reference(contigs)
        got = c.solve_compare_regions(batch, CompareConfig(enable_sequences=sequences, **okw), group_metrics=group_metrics)
        assert got.diff(want) == []
    assert ctx_pair[0].last_compare_was_one_shot() and not ctx_pair[1].last_compare_was_one_shot()
    return want


def test_known_answer_regions_with_sequences(ctx_pair, oracle):
    contigs, batch = scenarios.golden()
    check(ctx_pair, oracle, contigs, batch, sequences=True)
    check(ctx_pair, oracle, contigs, batch, sequences=False, group_metrics=False)


@pytest.mark.parametrize("seed,kw", [(301, {}), (302, {"repeat_unit": b"CA", "max_vars": 3}), (303, {"max_vars": 9, "max_len": 12}), (304, {"alphabet": b"ACGTN", "max_vars": 3})])
def test_fuzz_regions(ctx_pair, oracle, seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 3000, **kw)
    check(ctx_pair, oracle, contigs, batch)


def test_invalid_long_and_odd_inputs(ctx_pair, oracle):
    for sc in (scenarios.invalid_regions(), scenarios.long_allele_regions(), scenarios.non_acgt_regions(), scenarios.max_allele_regions()):
        check(ctx_pair, oracle, sc[0], sc[1], sequences=True)


def test_quota_and_autofail(ctx_pair, oracle):
    contigs, batch = scenarios.autofail_regions()
    check(ctx_pair, oracle, contigs, batch)
    contigs, batch = scenarios.quota_regions(3)
    check(ctx_pair, oracle, contigs, batch, max_branch_factor=3)


def test_benchmark_contig_pinned_and_pageable(ctx_pair, oracle):
    """one chr20-sized contig of the benchmark workload at full density, N-sprinkled: pageable arrays (bounce buffer) and pinned arrays (avk_host_alloc,
    direct DMA) give the same outputs; resident form too"""
    contig, batch = synth.config_indel_mix_v2(n_truth=int(synth.HG002_TRUTH_CALLS * synth.CHR20_LEN / sum(synth.GRCH38)), contig_len=synth.CHR20_LEN)
    contig = contig.copy()
    rng = np.random.default_rng(5)
    contig[rng.integers(0, contig.size, size=contig.size // 5000)] = ord("N")
    want = check(ctx_pair, oracle, [contig], batch, group_metrics=False)
    dev = ctx_pair[0]
    assert 0.8 * batch.n_regions < dev.last_lane_solved() < batch.n_regions
    import ctypes as C
    pb = dev.pinned_batch(batch)
    res = dev.pinned_results(pb)
    cb, cfg, ro = pb.c_struct(), CompareConfig(enable_sequences=False).c_struct(), res.c_struct()
    dev.set_option("emit_group_metrics", 0)
    for _ in range(2):
        res.status[:] = -7
        dev._check(dev.lib.avk_compare_batch(dev.handle, C.byref(cb), C.byref(cfg), C.byref(ro)))
        assert res.diff(want) == []
    dev.set_option("emit_group_metrics", 1)
    rb = dev.upload(batch)
    for _ in range(2):
        dev.compare_resident(rb, CompareConfig(enable_sequences=False))
    assert dev.download(rb, group_metrics=True).diff(oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS)) == []
    rb.free()


def test_pair_form(ctx_pair, oracle):
    for contigs, batch in (scenarios.fuzz_regions(331, 4000, max_vars=3, related=0.9), scenarios.invalid_regions(), scenarios.fuzz_regions(332, 2000, max_vars=7, related=0.95)):
        st_o, ex_o = oracle_lib.optimize_pairs(oracle, batch, contigs, threads=CPUS)
        for c in ctx_pair:
            c.upload_reference(contigs)
            st, ex = c.optimize_pairs(batch)
            assert np.array_equal(st_o, st) and np.array_equal(ex_o, ex)


def test_capacity_retry_of_a_device_packed_batch(oracle):
    """tiny workspaces: regions fail with CAPACITY on the device and are solved again by avk_results_download, which fetches the packed records of
    a device-packed batch for that"""
    import aardvark_amd
    ctx = aardvark_amd.Context(0)
    try:
        for k, v in dict(lds_bytes_per_wave=2048, lds2_bytes_per_wave=0, ws_bytes_per_wave=0, big_ws_bytes=4096).items():
            ctx.set_option(k, v)
        contigs, batch = scenarios.fuzz_regions(341, 400, max_vars=9, max_len=12)
        ctx.upload_reference(contigs)
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False))
        want = oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS)
        assert got.diff(want) == []
        # the per-label sums of a resident batch, taken after the download that repaired the regions, count the repaired regions too (one label on every region:
        # its sums are the batch's)
        rb = ctx.upload(batch)
        ctx.compare_resident(rb)
        res = ctx.download(rb, group_metrics=True)
        assert res.diff(want) == []
        n = batch.n_regions
        sums = ctx.label_tallies(rb, 1, np.arange(n + 1, dtype=np.uint64), np.zeros(n, np.uint32))
        assert np.array_equal(sums[0][:13 * 22], want.tally[:13 * 22])
        rb.free()
        ctx.set_option("capacity_retry", 0)
        starved = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False))
        assert (starved.status == 21).any()  # some regions do exhaust the last tier on the first try
        # the retry is gated on the kernels' own count of the regions they gave up on (last_tier_counts()[4]), not on a scan of the statuses: the two agree,
        # for the one-shot call, for a resident batch of either packer, and for either result form
        assert ctx.last_tier_counts()[4] == int((starved.status == 21).sum())
        for dev_pack in (1, 0):
            ctx.set_option("device_pack", dev_pack)
            rb = ctx.upload(batch)
            ctx.compare_resident(rb)
            for packed in (False, True):
                res = ctx.download(rb, group_metrics=False, packed=packed)
                st = res.status if not packed else res.expanded(ctx.lib, batch).status
                assert ctx.last_tier_counts()[4] == int((st == 21).sum()) > 0, (dev_pack, packed)
            rb.free()
    finally:
        ctx.close()


def test_batch_level_errors(ctx_pair):
    from aardvark_amd.api import AardvarkAmdError
    contigs, batch = scenarios.fuzz_regions(351, 50)
    bad = RegionBatch(batch.region_id, batch.contig_idx, batch.start, batch.end, batch.t_off, batch.t_cnt.copy(), batch.q_off, batch.q_cnt, batch.var_pos, batch.var_type,
                      batch.var_zyg, batch.var_raw_space, batch.a0_off, batch.a0_len, batch.a1_off, batch.a1_len, batch.allele_bytes)
    bad.t_cnt[7] = 0x80000000
    for c in ctx_pair:
        c.upload_reference(contigs)
        with pytest.raises(AardvarkAmdError, match="variant range"):
            c.solve_compare_regions(bad, CompareConfig(enable_sequences=False))
        assert c.solve_compare_regions(batch, CompareConfig(enable_sequences=False)).status.min() >= 0  # the context is fine afterwards


def test_packed_reference_off_with_lane_sized_batch(oracle):
    """use_packed_reference = 0 on a batch large enough for lane launches: the lanes (which read the 2-bit reference only) are not used"""
    import aardvark_amd
    contig, batch = synth.config_indel_mix_v2(n_truth=60_000, contig_len=24_000_000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
    for dp in (1, 0):
        ctx = aardvark_amd.Context(0)
        try:
            ctx.set_option("device_pack", dp)
            ctx.set_option("lane_min_batch", 0)
            ctx.set_option("lane_min_regions", 0)
            ctx.set_option("emit_group_metrics", 0)
            ctx.upload_reference([contig])
            rb = ctx.upload(batch)  # planned with the lanes on
            ctx.set_option("use_packed_reference", 0)
            ctx.compare_resident(rb, CompareConfig(enable_sequences=False))
            got = ctx.download(rb, group_metrics=False)
            assert got.diff(want) == []
            assert ctx.last_lane_solved() == 0
            rb.free()
        finally:
            ctx.close()


def test_batches_that_share_the_call_arrays(ctx_pair, oracle):
    """a job cut into batches of regions over ONE set of call arrays, results into ONE set of per-call output arrays (what compare_main.cpp does):
    a batch only writes the calls its regions own — in region order and, for the last case, with the batches' calls interleaved"""
    import ctypes as C
    from aardvark_amd._abi import ResultBatch
    contig, batch = synth.config_indel_mix_v2(n_truth=20_000, contig_len=8_000_000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
    n = batch.n_regions
    cfg = CompareConfig(enable_sequences=False).c_struct()
    for c in ctx_pair:
        c.upload_reference([contig])
        c.set_option("emit_group_metrics", 0)
        for parts in ([(0, n // 3), (n // 3, n // 2), (n // 2, n)], "interleaved"):
            whole = ResultBatch(batch, sequences=False, group_metrics=False)
            if parts == "interleaved":  # even regions, then odd regions: each batch's calls lie between the other's
                idx = [np.arange(0, n, 2), np.arange(1, n, 2)]
                subs = [RegionBatch(batch.region_id[i], batch.contig_idx[i], batch.start[i], batch.end[i], batch.t_off[i], batch.t_cnt[i], batch.q_off[i], batch.q_cnt[i], batch.var_pos,
                                    batch.var_type, batch.var_zyg, batch.var_raw_space, batch.a0_off, batch.a0_len, batch.a1_off, batch.a1_len, batch.allele_bytes) for i in idx]
            else:
                idx = [np.arange(lo, hi) for lo, hi in parts]
                subs = [batch.slice(lo, hi) for lo, hi in parts]
            for i, sub in zip(idx, subs):
                res = ResultBatch(sub, sequences=False, group_metrics=False)
                res.var_expected, res.var_observed, res.var_class, res.var_zyg = whole.var_expected, whole.var_observed, whole.var_class, whole.var_zyg
                cb, ro = sub.c_struct(), res.c_struct()
                c._check(c.lib.avk_compare_batch(c.handle, C.byref(cb), C.byref(cfg), C.byref(ro)))
                whole.status[i], whole.ed_h1[i], whole.ed_h2[i], whole.n_optima[i], whole.type_present[i] = res.status, res.ed_h1, res.ed_h2, res.n_optima, res.type_present
                whole.tally += res.tally
            assert whole.diff(want) == []
        c.set_option("emit_group_metrics", 1)


def test_a_shared_call_and_an_unowned_call_in_one_range(ctx_pair, oracle):
    """the counts of a batch's regions add up to the size of its call range, yet one call is owned by two regions and another by none: the unowned call keeps what
    the caller's arrays hold (ownership is counted on the device, DpIn::owned, not inferred from the sum)"""
    import ctypes as C
    from aardvark_amd._abi import ResultBatch
    contig, batch = synth.config_indel_mix_v2(n_truth=3_000, contig_len=1_200_000)
    # regions with exactly one truth call and one query call each, far apart in the list
    ones = np.nonzero((batch.t_cnt == 1) & (batch.q_cnt == 1))[0]
    r0, r1 = int(ones[3]), int(ones[len(ones) // 2])
    t_off = batch.t_off.copy()
    orphan = int(t_off[r1])
    t_off[r1] = t_off[r0]  # region r1's truth call is now region r0's: shared; its own is nobody's
    odd = RegionBatch(batch.region_id, batch.contig_idx, batch.start, batch.end, t_off, batch.t_cnt, batch.q_off, batch.q_cnt, batch.var_pos, batch.var_type, batch.var_zyg,
                      batch.var_raw_space, batch.a0_off, batch.a0_len, batch.a1_off, batch.a1_len, batch.allele_bytes)
    cfg = CompareConfig(enable_sequences=False).c_struct()
    for c in ctx_pair:
        c.upload_reference([contig])
        res = ResultBatch(odd, sequences=False, group_metrics=False)
        for f in ("var_expected", "var_observed", "var_class", "var_zyg"):
            getattr(res, f)[:] = 0x5A
        cb, ro = odd.c_struct(), res.c_struct()
        c._check(c.lib.avk_compare_batch(c.handle, C.byref(cb), C.byref(cfg), C.byref(ro)))
        assert [int(getattr(res, f)[orphan]) for f in ("var_expected", "var_observed", "var_class", "var_zyg")] == [0x5A] * 4
        owned = np.ones(batch.n_variants, bool)
        owned[orphan] = False
        assert not (res.var_class[:batch.n_variants][owned] == 0x5A).any()
        # every region but r1 (whose window may not hold r0's call) solves as in the untouched batch
        want = oracle_lib.compare_batch(oracle, odd, [contig], threads=CPUS, group_metrics=False)
        assert np.array_equal(res.status, want.status) and np.array_equal(res.ed_h1, want.ed_h1)


def test_compact_form(ctx_pair, oracle):
    """avk_compare_compact (the batch in 20 + 17 bytes per region / call, widened on the device) == avk_compare_batch == oracle; pageable and pinned; with and
    without raw_allele_space; the resident form from a compact upload"""
    import ctypes as C
    from aardvark_amd import CompactBatch
    dev = ctx_pair[0]
    contig, batch = synth.config_indel_mix_v2(n_truth=40_000, contig_len=16_000_000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
    dev.upload_reference([contig])
    cb = CompactBatch.from_region_batch(batch)
    assert cb.var_raw_space is None and cb.nbytes() < 0.55 * sum(getattr(batch, f).nbytes for f in ("contig_idx", "start", "end", "t_off", "t_cnt", "q_off", "q_cnt", "var_pos", "var_type",
                                                                                                     "var_zyg", "var_raw_space", "a0_off", "a0_len", "a1_off", "a1_len", "allele_bytes"))
    assert dev.solve_compact(cb).diff(want) == []
    pc = dev.pinned_compact(cb)
    res = dev.pinned_results(pc)
    for _ in range(2):
        res.status[:] = -3
        assert dev.solve_compact(pc, res=res).diff(want) == []
    cb_raw = CompactBatch.from_region_batch(batch, keep_raw_space=True)
    assert dev.solve_compact(cb_raw).diff(want) == []
    # odd inputs in the compact form: invalid regions, other symbols, long alleles
    for sc in (scenarios.invalid_regions(), scenarios.non_acgt_regions(), scenarios.long_allele_regions(), scenarios.fuzz_regions(361, 1500, max_vars=7)):
        contigs, b = sc[0], sc[1]
        try:
            c2 = CompactBatch.from_region_batch(b)
        except ValueError:
            continue  # (a window that ends before it starts has no compact form)
        dev.upload_reference(contigs)
        assert dev.solve_compact(c2, res=None).diff(oracle_lib.compare_batch(oracle, b, contigs, threads=CPUS, group_metrics=False)) == []


def test_packed_form(ctx_pair, oracle):
    """avk_compare_packed (10 + 5 bytes per region / call and the allele bytes; offsets from two prefix sums on the device) == avk_compare_compact == oracle;
    pageable and pinned, with and without raw_allele_space and contig indices, BASEPAIR groups, several contigs, odd inputs, totals that do not add up"""
    import aardvark_amd
    from aardvark_amd import CompactBatch, PackedBatch
    dev = ctx_pair[0]
    contig, batch = synth.config_indel_mix_v2(n_truth=40_000, contig_len=16_000_000)
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
    dev.upload_reference([contig])
    cb = CompactBatch.from_region_batch(batch)
    pk = PackedBatch.from_compact(cb)
    assert pk.nbytes() < 0.5 * cb.nbytes()
    got = dev.solve_packed(pk)
    assert got.diff(want) == [] and np.array_equal(got.tally, want.tally)
    pp = dev.pinned_packed(pk)
    res = dev.pinned_results(pp, bp_groups=True)
    ref = dev.solve_compact(cb, res=aardvark_amd._abi.ResultBatch(cb, sequences=False, group_metrics=False, bp_groups=True))
    for _ in range(2):
        res.status[:] = -3
        assert dev.solve_packed(pp, res=res).diff(want) == []
        assert np.array_equal(res.bp_off, ref.bp_off) and np.array_equal(res.bp_groups[:res.bp_off[-1]], ref.bp_groups[:ref.bp_off[-1]])
    assert dev.solve_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch, keep_raw_space=True))).diff(want) == []
    # several contigs and more prefix-sum workgroups than one scan round (a quarter of a genome: 0.9 M regions, 2 M calls)
    contigs, b = synth.config_genome(scale=0.25)
    dev.upload_reference(contigs)
    c4 = CompactBatch.from_region_batch(b)
    assert dev.solve_packed(PackedBatch.from_compact(c4)).diff(dev.solve_compact(c4)) == []
    # odd inputs: invalid regions, other symbols, long alleles (over 255 bases: no packed form)
    n_packed = 0
    for sc in (scenarios.invalid_regions(), scenarios.non_acgt_regions(), scenarios.long_allele_regions(), scenarios.fuzz_regions(362, 1500, max_vars=7)):
        contigs, b = sc[0], sc[1]
        try:
            p2 = PackedBatch.from_compact(CompactBatch.from_region_batch(b))
        except ValueError:
            continue
        n_packed += 1
        dev.upload_reference(contigs)
        assert dev.solve_packed(p2).diff(oracle_lib.compare_batch(oracle, b, contigs, threads=CPUS, group_metrics=False)) == []
    assert n_packed >= 1
    # batches of no region, of one region, of a few regions (fewer pieces than streams and events of the upload)
    from aardvark_amd.dist import gather_calls
    dev.upload_reference([contig])
    for lo, hi in ((0, 0), (0, 1), (7, 10), (100, 164)):
        small = gather_calls(batch.slice(lo, hi))
        p3 = PackedBatch.from_compact(CompactBatch.from_region_batch(small))
        got3 = dev.solve_packed(p3)
        assert np.array_equal(got3.status, want.status[lo:hi]) and np.array_equal(got3.ed_h1, want.ed_h1[lo:hi]) and np.array_equal(got3.n_optima, want.n_optima[lo:hi]), (lo, hi)
        if hi > lo:
            assert got3.diff(oracle_lib.compare_batch(oracle, small, [contig], threads=CPUS, group_metrics=False)) == [], (lo, hi)
    # a batch whose counts do not add up to n_variants is refused, not read out of bounds
    bad = PackedBatch.from_compact(cb)
    bad.t_cnt = bad.t_cnt.copy()
    bad.t_cnt[5] += 1
    with pytest.raises(aardvark_amd.AardvarkAmdError):
        dev.upload_reference([contig])
        dev.solve_packed(bad)


def test_merge_of_three_call_sets_at_genome_density(oracle):
    """configs[4] at the size of one contig: three perturbed call sets of one chr20-sized contig at the benchmark's density through avk_merge_batch (pair
    expansion, pair solve and classification on the device) against oracle pairs + the restated rule; every strategy on a slice"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import merge_oracle as mo
    import aardvark_amd
    from aardvark_amd.merge import MergeConfig, MultiBatch, PackedMultiBatch, merge_multi_batch, pair_batch, pinned_multi_batch
    n_truth = int(synth.HG002_TRUTH_CALLS * synth.CHR20_LEN / sum(synth.GRCH38))
    contig = synth.make_contig_fast(synth.CHR20_LEN, 20250103 + 19)
    rng = np.random.default_rng(20250103 + 119)
    bed = synth.make_bed(synth.CHR20_LEN, 1000, 0.9, rng)
    truth, info = synth.genome_truth(contig, bed, n_truth, 20250103 + 219)
    sets = [synth.genome_query(contig, bed, truth, info, seed, max(1, len(truth) // 100)) for seed in (20250105, 20250106, 20250107)]
    mb = MultiBatch(3, **synth.cluster_multi_v(contig, bed, sets, 50))

    def sub_of(b, nr):  # the first nr regions with their calls (the calls lie in region order)
        nvv = int(b.in_off[3 * nr]) if nr < b.n_regions else b.n_variants
        na = int(b.a0_off[nvv]) if nvv < b.n_variants else b.allele_bytes.size
        return MultiBatch(3, region_id=b.region_id[:nr], contig_idx=b.contig_idx[:nr], start=b.start[:nr], end=b.end[:nr], in_off=b.in_off[:3 * nr], in_cnt=b.in_cnt[:3 * nr],
                          var_pos=b.var_pos[:nvv], var_type=b.var_type[:nvv], var_zyg=b.var_zyg[:nvv], var_raw_space=b.var_raw_space[:nvv], a0_off=b.a0_off[:nvv], a0_len=b.a0_len[:nvv],
                          a1_off=b.a1_off[:nvv], a1_len=b.a1_len[:nvv], allele_bytes=b.allele_bytes[:na])

    ctx = aardvark_amd.Context(0)
    try:
        ctx.upload_reference([contig])
        got = merge_multi_batch(ctx, mb, MergeConfig(majority_voting_enabled=True))
        assert ctx.last_compare_was_one_shot()
        k, n = 3, mb.n_regions
        off, cnt = mb.in_off.reshape(n, k), mb.in_cnt.reshape(n, k)
        pairs = [(0, 1), (0, 2), (1, 2)]
        rep = lambda a: np.repeat(a, 3)
        pb = RegionBatch(rep(mb.region_id), rep(mb.contig_idx), rep(mb.start), rep(mb.end), np.stack([off[:, i] for i, _ in pairs], 1).reshape(-1),
                         np.stack([cnt[:, i] for i, _ in pairs], 1).reshape(-1), np.stack([off[:, j] for _, j in pairs], 1).reshape(-1), np.stack([cnt[:, j] for _, j in pairs], 1).reshape(-1),
                         mb.var_pos, mb.var_type, mb.var_zyg, mb.var_raw_space, mb.a0_off, mb.a0_len, mb.a1_off, mb.a1_len, mb.allele_bytes)
        st_o, ex_o = oracle_lib.optimize_pairs(oracle, pb, [contig], 50, threads=CPUS)
        ws, wc, wm = mo.classify_k3_majority(st_o, ex_o)
        assert np.array_equal(got.status, ws) and np.array_equal(got.classification, wc) and np.array_equal(got.members, wm)
        # the packed form of the same batch (avk_merge_packed: offsets by prefix sums on the device), from pageable and from pinned arrays
        pm = PackedMultiBatch.from_multi(mb)
        for form in (pm, pinned_multi_batch(ctx, pm), PackedMultiBatch.from_multi(mb, keep_raw_space=True)):
            got_k = merge_multi_batch(ctx, form, MergeConfig(majority_voting_enabled=True))
            assert ctx.last_compare_was_one_shot()
            assert np.array_equal(got_k.status, ws) and np.array_equal(got_k.classification, wc) and np.array_equal(got_k.members, wm)
        bad = PackedMultiBatch.from_multi(mb)
        bad.in_cnt = bad.in_cnt.copy()
        bad.in_cnt[11] += 1  # the counts no longer sum to n_variants: refused, not computed
        with pytest.raises(aardvark_amd.AardvarkAmdError):
            merge_multi_batch(ctx, bad, MergeConfig())
        # without the device packer the packed form is widened on the host and takes the wide path
        ctx.set_option("device_pack", 0)
        got_h = merge_multi_batch(ctx, PackedMultiBatch.from_multi(sub_of(mb, 2000)), MergeConfig(majority_voting_enabled=True))
        ctx.set_option("device_pack", 1)
        assert np.array_equal(got_h.status, ws[:2000]) and np.array_equal(got_h.classification, wc[:2000]) and np.array_equal(got_h.members, wm[:2000])
        # the same batch from pinned arrays (DMA instead of the bounce buffer)
        got_p = merge_multi_batch(ctx, pinned_multi_batch(ctx, mb), MergeConfig(majority_voting_enabled=True))
        assert np.array_equal(got_p.status, ws) and np.array_equal(got_p.classification, wc) and np.array_equal(got_p.members, wm)
        assert (wc == 1).sum() > 0.8 * n and (wc == 3).sum() > 0.01 * n
        # the general rule, region by region, for every strategy on a slice of the contig
        sub = MultiBatch(3, region_id=mb.region_id[:3000], contig_idx=mb.contig_idx[:3000], start=mb.start[:3000], end=mb.end[:3000], in_off=mb.in_off[:9000], in_cnt=mb.in_cnt[:9000],
                         var_pos=mb.var_pos, var_type=mb.var_type, var_zyg=mb.var_zyg, var_raw_space=mb.var_raw_space, a0_off=mb.a0_off, a0_len=mb.a0_len, a1_off=mb.a1_off, a1_len=mb.a1_len,
                         allele_bytes=mb.allele_bytes)
        for cfg in (MergeConfig(), MergeConfig(no_conflict_enabled=True), MergeConfig(no_conflict_enabled=True, majority_voting_enabled=True, conflict_selection=2)):
            dec = merge_multi_batch(ctx, sub, cfg).decoded()
            for m in range(sub.n_regions):
                if (st_o.reshape(-1, 3)[m] != 0).any():
                    continue
                w = mo.classify([int(x) for x in cnt[m]], lambda i, j: int(ex_o.reshape(-1, 3)[m][{(0, 1): 0, (0, 2): 1, (1, 2): 2}[(i, j)]]), cfg.no_conflict_enabled,
                                cfg.majority_voting_enabled, cfg.conflict_selection)
                assert dec[m][1] == tuple(w) or dec[m][1] == w, (m, dec[m], w)
    finally:
        ctx.close()


def test_two_lane_windows_with_different_starts_share_a_call(ctx_pair, oracle):
    """two overlapping windows of the lane classes, 30 bases apart, name the same truth call through explicit offsets: each region's fast record holds the call's
    position relative to ITS start (the packer's call slots are per call; the position is re-derived per region for forms with explicit offsets)"""
    rng = np.random.default_rng(5)
    contig = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20_000)
    n = 400
    starts = 200 + 45 * np.arange(n, dtype=np.uint64)  # windows of 100 bases that overlap their neighbours
    ends = starts + 100
    var_pos, var_type, var_zyg, a0, a1, alle = [], [], [], [], [], []
    t_off, q_off = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    for r in range(n):
        p = int(starts[r]) + 60  # inside window r at 60 and inside window r + 1 at 15
        ref = contig[p]
        alt = b"ACGT"[(b"ACGT".index(bytes([ref])) + 1) % 4]
        for side in range(2):  # the call once as a truth call, once as a query call
            (t_off if side == 0 else q_off)[r] = len(var_pos)
            var_pos.append(p), var_type.append(0), var_zyg.append(int(rng.integers(2, 6))), a0.append(1), a1.append(1)
            alle += [ref, alt]
    nv = len(var_pos)
    # odd regions take the truth call of their left neighbour (which their window holds at another offset) instead of their own
    t_off_shared = t_off.copy()
    t_off_shared[1::2] = t_off[0:-1:2]
    a0_off = 2 * np.arange(nv, dtype=np.uint64)
    batch = RegionBatch(np.arange(n, dtype=np.uint64), np.zeros(n, np.uint32), starts, ends, t_off_shared, np.ones(n, np.uint32), q_off, np.ones(n, np.uint32),
                        np.array(var_pos, np.uint64), np.array(var_type, np.uint8), np.array(var_zyg, np.uint8), np.ones(nv, np.uint32), a0_off, np.array(a0, np.uint32),
                        a0_off + 1, np.array(a1, np.uint32), np.array(alle, np.uint8))
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=CPUS, group_metrics=False)
    assert (want.status == 0).all() and (want.ed_h1[1::2] + want.ed_h2[1::2]).sum() > 0  # the shared calls are at other positions than the regions' own: edits
    for c in ctx_pair:
        c.upload_reference([contig])
        c.set_option("lane_min_regions", 0)
        c.set_option("lane_min_batch", 0)
        got = c.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
        assert np.array_equal(got.status, want.status) and np.array_equal(got.ed_h1, want.ed_h1) and np.array_equal(got.ed_h2, want.ed_h2)
        assert np.array_equal(got.tally, want.tally)
    assert ctx_pair[0].last_lane_solved() > n // 2
