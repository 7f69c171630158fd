"""Randomised schedules: none of the context options may change a result.  Every draw sets EVERY scheduling option of a fresh context to a random value of its table
(workspace budgets and slice sizes, block counts, thresholds of the lane classes, tile and head widths, kept node states, the quad / wide / lane kernels on and off,
LDS budgets of the wide kernel, static shares and claims, both packers) and solves one of four small workloads — region fuzz with up to three calls a side, het
clusters (large searches on small windows), a slice of the benchmark genome at full density with its repeat-run indels, windows of kilobases — through the one-shot
or the resident path, against the oracle, bit for bit.  A failing draw prints its seed and options.  (The round-4 slice-share race was found by an offline sweep
with non-default options; this is that sweep in the suite the driver runs.)"""
import os
import time

import numpy as np
import pytest

import oracle_lib
import scenarios
from aardvark_amd import CompareConfig, synth

pytestmark = pytest.mark.gpu
CPUS = min(os.cpu_count() or 1, 16)
N_DRAWS = int(os.environ.get("AVK_RANDOM_SCHEDULES", "240"))
SEED = int(os.environ.get("AVK_RANDOM_SEED", "20251003"))

# option -> values a draw picks from (the first one is the default)
TABLE = {
    "lds_bytes_per_wave": [10240, 4096, 20480, 0], "lds2_bytes_per_wave": [40960, 20480, 0], "lds_ed_cap": [48, 8, 200], "lds2_ed_cap": [48, 16],
    "ws_bytes_per_wave": [1 << 20, 1 << 18, 4 << 20], "big_ws_bytes": [64 << 20, 8 << 20, 256 << 20], "waves_per_cu": [12, 4, 32],
    "ws_budget_bytes": [96 << 30, 1 << 30, 2 << 30], "adaptive_ws": [1, 0],
    "solo_min_variants": [5, 3, 0, 9], "class_c_nodes_x2": [12, 1, 50, 1000],
    "lds_escalation": [1, 0], "static_pct": [75, 0, 100, 33], "claim": [2, 1, 7, 64],
    "wide_kernel": [1, 1, 0], "wide_lds_bytes": [16384, 8192, 65536, 24576], "wide_retry_lds_bytes": [65536, 0, 32768],
    "device_pack": [1, 1, 0], "use_packed_reference": [1, 1, 1, 0],
    "lane_kernel": [1, 1, 1, 0], "lane_min_regions": [0, 0, 2048], "lane_min_batch": [0, 0, 16384], "lane_width_one": [64, 32, 16, 8, 4], "lane_width_two": [64, 32, 16, 8, 4],
    "lane_width_three": [16, 8, 4, 32, 64], "lane_head_width": [16, 0, 4, 8, 32, 64], "hbm_ed_cap": [1024, 0, 8], "lane_pairs": [1, 0], "lane_node_cap": [32, 8, 250], "lane_quad": [1, 1, 0], "packed_source": [1, 1, 0], "kernel_copies": [1, 0, 2], "team_long_windows": [1, 1, 0, 2], "team_head_regions": [48, 2, 400],
}


@pytest.fixture(scope="module")
def workloads(oracle):
    from test_wide_parity import het_cluster_regions
    out = []
    contigs, batch = scenarios.fuzz_regions(901, 2500, max_vars=3, related=0.8)
    out.append(("region fuzz", contigs, batch))
    contigs, batch = het_cluster_regions(902, 500, n_sites=(2, 7), indel=0.2)
    out.append(("het clusters", contigs, batch))
    contig, bed, truth, query = synth.contig_calls(3, 6_000_000, 18_000 / 6_000_000, seed_ref=903, seed_query=904, str_frac=0.15, multi_frac=0.05)
    out.append(("genome slice", [contig], synth.cluster_regions_v(contig, bed, truth, query, 50)))
    contig, bed, truth, query = synth.contig_calls(5, 3_000_000, 3_800 / 3_000_000, seed_ref=905, seed_query=906, str_frac=0.15, multi_frac=0.05)  # the genome's density
    out.append(("large windows", [contig], synth.cluster_regions_v(contig, bed, truth, query, 1000)))
    def packed_form(batch):  # the batch as avk_packed_batch when it fits the form (a third of the draws hand it over that way: the packer then reads the packed arrays themselves)
        from aardvark_amd import CompactBatch, PackedBatch
        try:
            return PackedBatch.from_compact(CompactBatch.from_region_batch(batch))
        except ValueError:
            return None
    return [(name, contigs, batch, {gm: oracle_lib.compare_batch(oracle, batch, contigs, threads=CPUS, group_metrics=gm) for gm in (True, False)}, packed_form(batch)) for name, contigs, batch in out]


def test_random_option_sets_do_not_change_a_result(workloads):
    import aardvark_amd
    rng = np.random.default_rng(SEED)
    t0 = time.time()
    names = sorted(TABLE)
    n_lane = n_wide = 0
    for draw in range(N_DRAWS):
        opts = {k: TABLE[k][int(rng.integers(0, len(TABLE[k])))] for k in names}
        if opts["lds_bytes_per_wave"] == 0 and opts["lds2_bytes_per_wave"] == 0:
            opts["lds2_bytes_per_wave"] = 40960  # (the bulk needs one LDS tier)
        name, contigs, batch, want, pbatch = workloads[draw % len(workloads)]
        gm = bool(rng.integers(0, 2))
        resident = bool(rng.integers(0, 2))
        packed = bool(rng.integers(0, 2))
        as_packed_batch = pbatch is not None and int(rng.integers(0, 3)) == 0
        ctx = aardvark_amd.Context(0)
        try:
            for k in names:
                ctx.set_option(k, opts[k])
            ctx.set_option("emit_group_metrics", 1 if gm else 0)
            ctx.upload_reference(contigs)
            cfg = CompareConfig(enable_sequences=False)
            if as_packed_batch:
                from aardvark_amd import ResultBatch
                got = ctx.solve_packed(pbatch, cfg, res=ResultBatch(pbatch, sequences=False, group_metrics=gm, packed=packed))
            elif resident:
                rb = ctx.upload(batch)
                ctx.compare_resident(rb, cfg)
                got = ctx.download(rb, group_metrics=gm, packed=packed)
                rb.free()
            else:
                got = ctx.solve_compare_regions(batch, cfg, group_metrics=gm, packed=packed)
            diff = got.diff(want[gm])
            n_lane += ctx.last_lane_solved()
            n_wide += ctx.last_wide_solved()
        finally:
            ctx.close()
        assert diff == [], "draw %d (seed %d, %s, %s path, group metrics %s, packed %s): %s\noptions: %s" % (
            draw, SEED, name, "packed batch" if as_packed_batch else "resident" if resident else "one-shot", gm, packed, diff[:5], ",".join("%s=%d" % (k, opts[k]) for k in names))
    assert n_lane > 0 and n_wide > 0
    print("%d draws in %.0f s; regions through the lanes %d, through the wide kernel %d" % (N_DRAWS, time.time() - t0, n_lane, n_wide))
