"""Large parity sweep on the GPU box: many seeds x generator settings, every output array of the GPU path (through the C-ABI)
against the oracle.  Prints one line per case and a total; exits non-zero on the first difference (the failing case's seed and
settings are in the line).  BUDGET_S bounds the wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import aardvark_amd
from aardvark_amd import synth, CompareConfig
import oracle_lib
import scenarios

budget = float(os.environ.get("BUDGET_S", "600"))
seed_base = int(os.environ.get("SEED_BASE", "0"))  # other seeds than the committed reports used
lib = oracle_lib.load()
ctx = aardvark_amd.Context(0)
ctx.set_option("lane_min_regions", 0)  # every class of the lane-per-region kernel, whatever the batch size
for kv in os.environ.get("AVK_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
t_start = time.time()
total = 0
cases = 0


def check(name, contigs, batch, sequences=True, mbf=50):
    global total, cases
    t0 = time.time()
    want = oracle_lib.compare_batch(lib, batch, contigs, sequences=sequences, threads=min(16, os.cpu_count() or 1), max_branch_factor=mbf)
    t1 = time.time()
    ctx.upload_reference(contigs)
    got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=sequences, max_branch_factor=mbf), packed=True)  # the wide arrays and the packed result form
    t2 = time.time()
    d = got.diff(want)
    lanes_n, wide_n, tiers_n = ctx.last_lane_solved(), ctx.last_wide_solved(), ctx.last_tier_counts()
    if not d:
        d = ["packed:" + x for x in got.expanded(ctx.lib, batch).diff(want)]
    one_shot = " one-shot" if ctx.last_compare_was_one_shot() else ""
    if not d and os.environ.get("FORMS", "1") != "0":  # the same batch through the compact and the packed form (avk_compare_compact / avk_compare_packed), where it fits them
        from aardvark_amd import CompactBatch, PackedBatch
        from aardvark_amd.dist import gather_calls
        try:
            own = gather_calls(batch)  # calls region by region: the layout the narrow forms imply
            cb = CompactBatch.from_region_batch(own)
            want_own = oracle_lib.compare_batch(lib, own, contigs, sequences=False, threads=min(16, os.cpu_count() or 1), max_branch_factor=mbf, group_metrics=False)
            cfg = CompareConfig(enable_sequences=False, max_branch_factor=mbf)
            d = ["compact:" + x for x in ctx.solve_compact(cb, cfg).diff(want_own)]
            one_shot += " +compact"
            try:
                pk = PackedBatch.from_compact(cb)
                d += ["packed:" + x for x in ctx.solve_packed(pk, cfg).diff(want_own)]
                one_shot += "+packed"
            except ValueError:
                pass
        except ValueError:
            pass
    total += batch.n_regions
    cases += 1
    print("%-70s %8d regions  oracle %.2fs gpu %.2fs  tiers %s lanes %d wide %d%s  status!=0: %d  %s" % (
        name, batch.n_regions, t1 - t0, t2 - t1, tiers_n, lanes_n, wide_n, one_shot,
        int((want.status != 0).sum()), "OK" if not d else "DIFF " + str(d)), flush=True)
    if d:
        n = batch.n_regions
        bad = np.nonzero((got.status != want.status) | (got.ed_h1 != want.ed_h1) | (got.ed_h2 != want.ed_h2) |
                         (got.group_metrics.reshape(n, -1) != want.group_metrics.reshape(n, -1)).any(axis=1))[0]
        print("bad regions:", bad[:20].tolist(), len(bad))
        sys.exit(1)


FUZZ = [dict(), dict(repeat_unit=b"CAG", max_vars=6), dict(max_len=20, span=(30, 260)), dict(max_vars=9, span=(40, 200)), dict(related=0.95, max_vars=7),
        dict(repeat_unit=b"A", max_vars=5, max_len=12), dict(repeat_unit=b"AT", max_vars=8, span=(20, 120)), dict(alphabet=b"AC", max_vars=6),
        dict(max_vars=12, span=(60, 300), related=0.9), dict(max_len=40, span=(100, 400), max_vars=4), dict(alphabet=b"ACGTNacgt", max_vars=5),
        dict(max_vars=3, span=(12, 40)), dict(repeat_unit=b"GGC", max_vars=10, span=(50, 250), related=0.9),
        dict(max_vars=2), dict(max_vars=2, repeat_unit=b"CA", related=0.9), dict(max_vars=2, max_len=16, span=(20, 190)), dict(max_vars=2, repeat_unit=b"A", max_len=5),
        dict(max_vars=2, alphabet=b"ACGT" * 50 + b"Nc")]
rnd = 0
while time.time() - t_start < budget:
    for i, kw in enumerate(FUZZ):
        if time.time() - t_start > budget:
            break
        seed = seed_base + 1000 + 100 * rnd + i
        contigs, batch = scenarios.fuzz_regions(seed, int(os.environ.get("FUZZ_N", "20000")), **kw)
        check("fuzz seed %d %s" % (seed, kw), contigs, batch, mbf=(50, 50, 7, 2)[rnd % 4], sequences=(rnd + i) % 2 == 0)  # without sequences: the lane kernel takes its classes
    # call-set shaped cases: SNV + indel truth, queries perturbed at several error levels
    for j, (drop, flip, change, extra) in enumerate(((0.01, 0.005, 0.005, 2000), (0.1, 0.05, 0.05, 20000), (0.3, 0.2, 0.2, 60000))):
        if time.time() - t_start > budget:
            break
        seed = seed_base + 5000 + 10 * rnd + j
        length = 20_000_000
        contig = synth.make_contig(length, seed)
        rng = np.random.default_rng(seed + 1)
        bed = synth.make_bed(length, 300, 0.9, rng)
        truth = synth.indel_truth(contig, bed, 150_000, seed + 2, snv_frac=(0.82, 0.6, 0.4)[j], close_frac=(0.03, 0.1, 0.2)[j], str_frac=(0.05, 0.1, 0.15)[j])
        if rnd % 2 == 1:  # multi-allelic sites (two calls at one position), in the truth set and — perturbed like the rest — in the query
            truth = synth.add_multiallelic(truth, 0.02 * (j + 1), seed + 5)
        query = synth.perturb_query(contig, bed, truth, seed + 3, extra, drop, flip, change)
        batch = synth.cluster_regions(length, bed, truth, query, (50, 20, 120)[rnd % 3])
        check("callset seed %d drop %.2f flip %.2f change %.2f extra %d gap %d%s" % (seed, drop, flip, change, extra, (50, 20, 120)[rnd % 3],
                                                                                 " multiallelic" if rnd % 2 == 1 else ""), [contig], batch,
              sequences=(rnd % 2 == 0))
    # the benchmark workload's generator (multi-allelic sites, repeat-run indels at shifted positions) at three sizes: the largest goes
    # through the one-shot path of avk_compare_batch when the per-region blocks are off
    for j, (n_truth, length) in enumerate(((20_000, 8_000_000), (80_000, 30_000_000), (160_000, 64_000_000))):
        if time.time() - t_start > budget:
            break
        seed = seed_base + 9000 + 10 * rnd + j
        contig, bed, truth, query = synth.contig_calls(j, length, n_truth / length, seed_ref=seed, seed_query=seed + 1, str_frac=(0.05, 0.2, 0.1)[j], multi_frac=(0.02, 0.1, 0.05)[j])
        batch = synth.cluster_regions_v(contig, bed, truth, query, (50, 20, 120)[rnd % 3])
        ctx.set_option("emit_group_metrics", 0 if j == 2 else 1)
        if j == 2:
            want = oracle_lib.compare_batch(lib, batch, [contig], threads=min(16, os.cpu_count() or 1), group_metrics=False)
            ctx.upload_reference([contig])
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False)
            d = got.diff(want)
            total += batch.n_regions
            cases += 1
            print("%-70s %8d regions  lanes %d%s  %s" % ("workload seed %d (tally-only outputs)" % seed, batch.n_regions, ctx.last_lane_solved(),
                                                       " one-shot" if ctx.last_compare_was_one_shot() else "", "OK" if not d else "DIFF " + str(d)), flush=True)
            if d:
                sys.exit(1)
        else:
            check("workload seed %d n_truth %d" % (seed, n_truth), [contig], batch, sequences=False)
        ctx.set_option("emit_group_metrics", 1)
    rnd += 1
print("ALL OK: %d cases, %d regions in %.0f s" % (cases, total, time.time() - t_start))
