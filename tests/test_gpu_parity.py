"""Parity tests proper: the HIP kernels on a real MI355X, called through the C-ABI
(libaardvark_amd.so), against the CPU oracle — bit-exact on every output array — plus
size-independent properties at the benchmark's full size."""
import os

import numpy as np
import pytest

import oracle_lib
import scenarios

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import aardvark_amd
    import torch
    torch.cuda.init()  # torch ships its own HIP runtime: it must come up before the library's (one test hands the library a torch tensor)
    c = aardvark_amd.Context(0)
    yield c
    c.close()


def run(ctx, contigs, batch, sequences=True, max_branch_factor=50):
    from aardvark_amd import CompareConfig
    ctx.upload_reference(contigs)
    return ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=sequences, max_branch_factor=max_branch_factor))


def check(ctx, oracle, contigs, batch, **kw):
    want = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True, threads=min(os.cpu_count() or 1, 32), **kw)
    got = run(ctx, contigs, batch, **kw)
    assert got.diff(want) == []
    return got, want


def test_native_library_is_the_one_running(ctx):
    import aardvark_amd
    assert os.path.exists(aardvark_amd.library_path())
    maps = open("/proc/self/maps").read()
    assert os.path.basename(aardvark_amd.library_path()) in maps


def test_reference_known_answer_regions(ctx, oracle):
    contigs, batch = scenarios.golden()
    got, _ = check(ctx, oracle, contigs, batch)
    assert ctx.last_tier_counts()[0] == batch.n_regions


def test_chr20_snv_full_size(ctx, oracle):
    """BASELINE.json configs[1] at full size: every region, variant decision and tally"""
    from aardvark_amd import synth
    contig, batch = synth.config_chr20_snv()
    got, want = check(ctx, oracle, [contig], batch)
    # size-independent properties
    assert np.array_equal(got.tally[:-2], got.group_metrics.astype(np.uint64).sum(axis=0).reshape(-1))  # checksum of checksums
    assert int(got.tally[-2]) == batch.n_regions and int(got.tally[-1]) == 0
    again = run(ctx, [contig], batch)
    assert again.diff(got) == []  # idempotent
    # region order does not matter: reversed batch -> reversed results, same tally
    rev = batch.slice(0, batch.n_regions)
    for f in ("region_id", "contig_idx", "start", "end", "t_off", "t_cnt", "q_off", "q_cnt"):
        setattr(rev, f, np.ascontiguousarray(getattr(batch, f)[::-1]))
    got_rev = run(ctx, [contig], rev, sequences=False)
    assert np.array_equal(got_rev.status[::-1], got.status) and np.array_equal(got_rev.group_metrics[::-1], got.group_metrics)
    assert np.array_equal(got_rev.var_class, got.var_class) and np.array_equal(got_rev.tally, got.tally)


def test_indel_mix(ctx, oracle):
    contigs, batch = scenarios.indel_small(20000)
    check(ctx, oracle, contigs, batch)


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, {"repeat_unit": b"CAG", "max_vars": 6}), (13, {"max_len": 20, "span": (30, 260)}),
                                      (14, {"max_vars": 9, "span": (40, 200)})])
def test_fuzz_regions(ctx, oracle, seed, kw):
    contigs, batch = scenarios.fuzz_regions(seed, 1500, **kw)
    check(ctx, oracle, contigs, batch)


def test_non_acgt_bytes(ctx, oracle):
    contigs, batch = scenarios.non_acgt_regions()
    check(ctx, oracle, contigs, batch)


def test_branch_quota_decides(ctx, oracle):
    contigs, batch = scenarios.quota_regions(3, n=6, n_query=14)
    check(ctx, oracle, contigs, batch)
    check(ctx, oracle, contigs, batch, max_branch_factor=5)
    check(ctx, oracle, contigs, batch, max_branch_factor=1)


def test_auto_fail_pruning(ctx, oracle):
    contigs, batch = scenarios.autofail_regions()
    check(ctx, oracle, contigs, batch)
    assert oracle_lib.stats(oracle)["max_pops_b"] > 500


def test_invalid_and_degenerate_inputs(ctx, oracle):
    contigs, batch = scenarios.invalid_regions()
    got, want = check(ctx, oracle, contigs, batch)
    assert sorted(set(want.status.tolist())) == [0, 6, 20]
    assert int(got.tally[-1]) == int((want.status != 0).sum())


def test_empty_batch(ctx):
    from aardvark_amd import RegionBatch
    contigs, _ = scenarios.golden()
    batch = RegionBatch.from_regions([])
    got = run(ctx, contigs, batch)
    assert got.status.size == 0 and int(got.tally.sum()) == 0


def test_long_alleles_and_large_edit_distance(ctx, oracle):
    """3 kbp alleles, edit distances in the thousands: wavefronts live in the HBM tiers"""
    contigs, batch = scenarios.long_allele_regions()
    check(ctx, oracle, contigs, batch)
    assert ctx.last_tier_counts()[0] == 0 and sum(ctx.last_tier_counts()) == batch.n_regions


def test_results_do_not_depend_on_the_workspace_tier(oracle):
    import aardvark_amd
    contigs, batch = scenarios.fuzz_regions(21, 800)
    want = oracle_lib.compare_batch(oracle, batch, contigs, sequences=True, threads=8)
    for opts in ({"lds_bytes_per_wave": 0}, {"lds_bytes_per_wave": 0, "lds2_bytes_per_wave": 0},
                 {"lds_bytes_per_wave": 4096, "lds_ed_cap": 2, "lds2_bytes_per_wave": 6144, "lds2_ed_cap": 4},
                 {"lds_bytes_per_wave": 0, "lds2_bytes_per_wave": 0, "ws_bytes_per_wave": 0},
                 {"lds_bytes_per_wave": 32768, "lds_ed_cap": 64, "waves_per_cu": 4},
                 # scheduling variants: no solo launches, no in-workgroup escalation, the four-launch chain, solo waves with
                 # several regions each, tiny tier-0 slices (every other region escalates inside its workgroup), tier-2
                 # slices too small (big slices claimed in place)
                 {"solo_min_variants": 0}, {"lds_escalation": 0}, {"lds_escalation": 0, "solo_min_variants": 0},
                 {"solo_min_variants": 3, "class_c_nodes_x2": 3},
                 {"lds_bytes_per_wave": 4096, "solo_min_variants": 0},
                 {"lds_bytes_per_wave": 4096, "lds2_bytes_per_wave": 8192, "ws_bytes_per_wave": 12288},
                 # launch-chain variants: work dealt only statically / only by claims of 1 and 7, the bulk held back for the side
                 # streams, no timing events, a class C list long enough to be shared with the main stream's HBM launch
                 {"static_pct": 100}, {"static_pct": 0, "claim": 1}, {"static_pct": 10, "claim": 7},
                 {"class_c_nodes_x2": 1000, "solo_min_variants": 2}):
        c = aardvark_amd.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        got = run(c, contigs, batch)
        assert got.diff(want) == [], opts
        tiers = c.last_tier_counts()
        if opts.get("lds_bytes_per_wave") == 0:
            assert tiers[0] == 0
        c.close()
    c = aardvark_amd.Context(0)
    c.set_option("lds_bytes_per_wave", 2048)
    c.set_option("lds2_bytes_per_wave", 0)
    c.set_option("ws_bytes_per_wave", 0)
    c.set_option("big_ws_bytes", 4096)
    c.set_option("capacity_retry", 0)
    starved = run(c, contigs, batch)
    assert set(starved.status.tolist()) <= {0, 21} and (starved.status == 21).any()
    ok = starved.status == 0
    assert np.array_equal(starved.group_metrics[ok], want.group_metrics[ok])
    # the library solves what exhausted the last tier again, in larger slices: no region of the caller's batch stays a capacity failure
    c.set_option("capacity_retry", 1)
    retried = run(c, contigs, batch)
    assert retried.diff(want) == []
    c.close()


def test_resident_path_and_device_tally(ctx, oracle):
    """upload once, run twice from HBM, tally delivered into a caller-owned device buffer"""
    import torch
    from aardvark_amd import CompareConfig, TALLY_LEN
    contigs, batch = scenarios.chr20_small(3000)
    ctx.upload_reference(contigs)
    rb = ctx.upload(batch)
    tally = torch.zeros(TALLY_LEN, dtype=torch.int64, device="cuda:0")
    for _ in range(2):
        ctx.compare_resident(rb, CompareConfig(enable_sequences=False), tally.data_ptr())
    ctx.synchronize()
    got = ctx.download(rb)
    want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8)
    assert got.diff(want) == []
    assert np.array_equal(tally.cpu().numpy().astype(np.uint64), want.tally)
    assert ctx.last_kernel_ms() > 0
    rb.free()
