"""Kernel-logic parity on CPU: the HIP solver source (aardvark_amd/csrc/avk_solver.inl) executed
by the lane emulator (tests/emu) against the oracle, bit for bit, on every scenario.  These run
in the GPU-less container; the same scenarios run on the real kernels in test_gpu_parity.py."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import scenarios

EMU_THREADS = 8


def check(oracle, contigs, batch, **emu_kw):
    want = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True, threads=4)
    got = emu_lib.compare_batch(batch, contigs, sequences=True, threads=EMU_THREADS, **emu_kw)
    assert got.diff(want) == []
    return got, want


def test_tally_without_per_region_blocks(oracle):
    """emit_group_metrics = 0 (what the benchmark runs): only the metric groups of the types that occur are cleared and
    flushed per region; the batch tally and the per-variant decisions must not change"""
    for contigs, batch in (scenarios.golden(), scenarios.fuzz_regions(17, 150), scenarios.indel_small(1500), scenarios.chr20_small(800)):
        want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
        got = emu_lib.compare_batch(batch, contigs, threads=EMU_THREADS, group_metrics=False)
        want.group_metrics = None
        assert got.diff(want) == []
        assert np.array_equal(got.tally, want.tally)


def test_reference_known_answer_regions(oracle):
    contigs, batch = scenarios.golden()
    got, _ = check(oracle, contigs, batch, n_waves=2)
    assert got.tier_counts[0] + got.tier_counts[1] == batch.n_regions  # all of them fit the LDS tiers (tier 1 = solo waves)


def test_chr20_snv_regions(oracle):
    contigs, batch = scenarios.chr20_small(3000)
    check(oracle, contigs, batch, n_waves=16)


def test_indel_mix_regions(oracle):
    contigs, batch = scenarios.indel_small(1500)
    check(oracle, contigs, batch, n_waves=16)


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, {"repeat_unit": b"CAG", "max_vars": 6}), (13, {"max_len": 20, "span": (30, 260)})])
def test_fuzz_regions(oracle, seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 150, **kw)
    check(oracle, contigs, batch, n_waves=16)


def test_non_acgt_bytes(oracle):
    contigs, batch = scenarios.non_acgt_regions()
    check(oracle, contigs, batch, n_waves=8)


def test_branch_quota_decides(oracle):
    """more orientations than max_branch_factor: pop order under the quota must be the reference's"""
    contigs, batch = scenarios.quota_regions(3, n=3, n_query=12)
    got, want = check(oracle, contigs, batch, n_waves=4)
    assert oracle_lib.stats(oracle)["max_optima"] >= 1
    # a smaller quota changes the work but not the agreement
    want5 = oracle_lib.compare_batch(oracle, batch, contigs, max_branch_factor=5, sequences=True)
    got5 = emu_lib.compare_batch(batch, contigs, max_branch_factor=5, sequences=True, n_waves=4, threads=EMU_THREADS)
    assert got5.diff(want5) == []


def test_auto_fail_pruning(oracle):
    contigs, batch = scenarios.autofail_regions()
    check(oracle, contigs, batch, n_waves=4)
    assert oracle_lib.stats(oracle)["max_pops_b"] > 500  # the scenario really reaches the 500-pop rule


def test_invalid_and_degenerate_inputs(oracle):
    contigs, batch = scenarios.invalid_regions()
    got, want = check(oracle, contigs, batch, n_waves=2)
    assert sorted(set(want.status.tolist())) == [0, 6, 20]
    assert int(got.tally[-1]) == int((want.status != 0).sum())  # error_blocks
    assert int(got.tally[-2]) == int((want.status == 0).sum())  # solved_blocks


def test_results_do_not_depend_on_the_workspace_tier(oracle):
    """the same regions solved in the LDS slice, in the per-wave HBM slice and in the overflow
    pass give identical outputs; a workspace that is too small everywhere reports CAPACITY"""
    contigs, batch = scenarios.fuzz_regions(21, 60)
    want = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True)
    lds_only = emu_lib.compare_batch(batch, contigs, sequences=True, lds_bytes=64 * 1024, lds_ed_cap=64, lds2_bytes=0, ws_bytes=0, big_ws_bytes=8 << 20, threads=EMU_THREADS)
    lds2_only = emu_lib.compare_batch(batch, contigs, sequences=True, lds_bytes=0, lds2_bytes=40 * 1024, ws_bytes=0, big_ws_bytes=8 << 20, threads=EMU_THREADS)
    hbm_only = emu_lib.compare_batch(batch, contigs, sequences=True, lds_bytes=0, lds2_bytes=0, ws_bytes=1 << 20, big_ws_bytes=8 << 20, threads=EMU_THREADS)
    big_only = emu_lib.compare_batch(batch, contigs, sequences=True, lds_bytes=0, lds2_bytes=0, ws_bytes=0, big_ws_bytes=8 << 20, threads=EMU_THREADS)
    tiny = dict(sequences=True, lds_bytes=4096, lds_ed_cap=2, lds2_bytes=6144, lds2_ed_cap=4, ws_bytes=1 << 20, big_ws_bytes=8 << 20, threads=EMU_THREADS)
    tiny_lds = emu_lib.compare_batch(batch, contigs, lds2_overflow_pass=1, solo_min_variants=0, **tiny)  # four-launch chain
    tiny_solo = emu_lib.compare_batch(batch, contigs, solo_min_variants=3, **dict(tiny, lds2_bytes=40 * 1024, lds2_ed_cap=48))  # solo waves, overflow straight to the HBM tier
    tiny.update(ws_bytes=12 * 1024)
    tiny_hbm = emu_lib.compare_batch(batch, contigs, **tiny)  # tier-2 slices too small: big slices claimed in place
    # in-workgroup escalation: a wave that outgrows its 4 KB slice takes its workgroup's 16 KB while the siblings park
    tiny_esc = emu_lib.compare_batch(batch, contigs, sequences=True, lds_bytes=4096, lds_ed_cap=48, solo_min_variants=0, n_waves=8, threads=EMU_THREADS)
    tiny_noesc = emu_lib.compare_batch(batch, contigs, sequences=True, lds_bytes=4096, lds_ed_cap=48, solo_min_variants=0, lds_escalation=0, threads=EMU_THREADS)
    assert tiny_esc.tier_counts[1] > 0 and tiny_noesc.tier_counts[1] == 0
    for got in (lds_only, lds2_only, hbm_only, big_only, tiny_lds, tiny_solo, tiny_hbm, tiny_esc, tiny_noesc):
        assert got.diff(want) == []
    assert hbm_only.tier_counts[0] == 0 and hbm_only.tier_counts[1] == 0 and big_only.tier_counts[3] == batch.n_regions
    assert tiny_lds.tier_counts[1] > 0 and tiny_lds.tier_counts[2] > 0 and sum(tiny_lds.tier_counts) == batch.n_regions
    assert tiny_solo.tier_counts[1] > 0 and tiny_solo.tier_counts[2] > 0 and sum(tiny_solo.tier_counts) == batch.n_regions
    assert tiny_hbm.tier_counts[3] > 0 and tiny_hbm.tier_counts[4] == 0 and sum(tiny_hbm.tier_counts) == batch.n_regions
    starved = emu_lib.compare_batch(batch, contigs, lds_bytes=2048, lds_ed_cap=2, lds2_bytes=0, ws_bytes=0, big_ws_bytes=4096, threads=EMU_THREADS)
    assert set(starved.status.tolist()) <= {0, 21} and (starved.status == 21).any()
    assert starved.tier_counts[4] == int((starved.status == 21).sum())
    ok = starved.status == 0
    assert np.array_equal(starved.ed_h1[ok], want.ed_h1[ok]) and np.array_equal(starved.group_metrics[ok], want.group_metrics[ok])


@pytest.mark.timeout(600)
def test_long_alleles_and_large_edit_distance(oracle):
    contigs, batch = scenarios.long_allele_regions()
    sub = batch.slice(2, 4)  # the 600 bp SV pair and the TR pair (the 3 kbp pair runs on the GPU test)
    check(oracle, contigs, sub, n_waves=2)


def test_hidden_exact_shortcut(oracle):
    """--enable-exact-shortcut (waffle_solver.rs:171-199, :534-601): exact regions take the shortcut
    metrics, the others the full path"""
    for contigs, batch in (scenarios.golden(), scenarios.chr20_small(1200), scenarios.fuzz_regions(31, 120)):
        want = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True, exact_shortcut=True, threads=4)
        got = emu_lib.compare_batch(batch, contigs, sequences=True, exact_shortcut=True, threads=EMU_THREADS)
        assert got.diff(want) == []
        plain = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True, threads=4)
        assert want.diff(plain) != []  # the shortcut really changes the metrics (no RECORD_BP, fewer type entries)


def test_class_c_list_shared_between_the_hbm_launches(oracle, monkeypatch):
    """the records predicted to outgrow tier 1 are taken ticket by ticket from one counter by the HBM solo launch and by the main
    stream's HBM launch: every split of the list between the two gives the same results (here: all to one, all to the other),
    at thresholds that put almost nothing / a fifth of the batch into the list"""
    contigs, batch = scenarios.indel_small(4000)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8)
    seen = set()
    for skip in (False, True):
        for c in ("6", "12", "40"):
            monkeypatch.setenv("AVK_EMU_CLASS_C", c)
            if skip:
                monkeypatch.setenv("AVK_EMU_SKIP_HBM_SOLO", "1")
            got = emu_lib.compare_batch(batch, contigs, threads=8, lane_kernel=False, wide_kernel=False)  # every region through the wave-per-region launches
            assert got.diff(want) == []
            seen.add(got.tier_counts[2])
    assert max(seen) >= 50 and min(seen) < 20  # (regions of the lane-per-region classes are never predicted into the HBM list)


def test_large_windows_with_many_calls(oracle):
    """--min-variant-gap 1000 (windows of kilobases, a dozen and more calls per region): the searches run in the HBM tier, whose node records are large enough
    for the copies of their used part only (hap_copy_used) and whose wavefronts are sized by the region's own bound (hbm_ed_cap)"""
    import ctypes as C
    from aardvark_amd import synth
    from aardvark_amd.dist import gather_calls, take_regions
    contigs, batch = synth.config_genome(scale=0.0002, threads=2, gap=1000)
    calls = batch.t_cnt.astype(np.int64) + batch.q_cnt
    keep = np.union1d(np.argsort(calls)[-6:], np.arange(0, batch.n_regions, 12))  # the six regions with most calls and a twelfth of the rest: the emulator takes seconds for the largest
    batch = gather_calls(take_regions(batch, keep))
    assert batch.n_regions >= 20 and int((batch.end - batch.start).max()) > 2000 and int((batch.t_cnt + batch.q_cnt).max()) >= 12
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4)
    lib = emu_lib.load()
    lib.emu_set_hbm_ed_cap.argtypes = [C.c_uint32]
    try:
        for cap in (1024, 8):  # (8: most regions' bounds are above it — those keep two entries per base, as with 0)
            lib.emu_set_hbm_ed_cap(cap)
            got = emu_lib.compare_batch(batch, contigs, threads=EMU_THREADS, lane_kernel=False, big_ws_bytes=512 << 20)  # (one region of 52 calls in 11 kbp wants 300 MB)
            assert got.diff(want) == [], cap
            assert got.tier_counts[2] + got.tier_counts[3] > 0.5 * batch.n_regions
    finally:
        lib.emu_set_hbm_ed_cap(1024)
