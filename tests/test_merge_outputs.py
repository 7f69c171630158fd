"""`merge` end to end: region generation over k VCFs (avf_feed_merge), the classification of every region
(pairs from the oracle on CPU / avk_merge_batch on the GPU), and the writers of libaardvark_feeder.so — passing.vcf.gz,
regions.bed.gz, failed_regions.bed.gz with their tabix indexes and the merge summary table — against the Python restatements
(oracle/feeder_oracle.py, oracle/merge_oracle.py)."""
import gzip
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import feeder_oracle as fo  # noqa: E402
import merge_oracle as mo  # noqa: E402
import oracle_lib  # noqa: E402
from aardvark_amd import feeder, synth  # noqa: E402
from aardvark_amd._abi import VT, ZYG  # noqa: E402
from aardvark_amd.merge import MergeConfig, MergeResult, MultiBatch, pair_batch, solve_merge_regions  # noqa: E402
from test_feeder import ZNAME, bgzf_blocks, reg2bins, vcf_text, write_text  # noqa: E402

CLS_CODE = {"different": 0, "identical": 1, "no_conflict": 2, "majority": 3, "conflict_select": 4}
CONFIGS = [dict(), dict(no_conflict_enabled=True), dict(majority_voting_enabled=True), dict(conflict_selection=1),
           dict(no_conflict_enabled=True, majority_voting_enabled=True, conflict_selection=2)]


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def write_case(tmp_path, n_truth=400, length=200_000, k=4):
    """k call sets of one sample: input 0 = the base set, the others perturbed copies of it (one of them only slightly), over two
    contigs' worth of BED intervals on one contig"""
    contig = synth.make_contig(length, 31)
    rng = np.random.default_rng(32)
    bed = synth.make_bed(length, 12, 0.9, rng)
    base = synth.indel_truth(contig, bed, n_truth, 33)
    sets = [base] + [synth.perturb_query(contig, bed, base, 40 + i, (300, 30, 8)[min(i - 1, 2)]) for i in range(1, k)]
    p = {"fa": str(tmp_path / "m.fa"), "bed": str(tmp_path / "m.bed"), "out": str(tmp_path / "merged"), "vcfs": []}
    seq = bytes(contig).decode()
    write_text(p["fa"], ">chrM1\n" + "\n".join(seq[i:i + 70] for i in range(0, length, 70)) + "\n")
    write_text(p["bed"], "".join("chrM1\t%d\t%d\n" % (a, b) for a, b in bed))
    for i, cs in enumerate(sets):
        path = str(tmp_path / ("in%d.vcf.gz" % i))
        write_text(path, vcf_text("chrM1", [(int(cs.pos[j]), cs.ref[j].decode(), cs.alt[j].decode(), ZNAME[int(cs.zyg[j])]) for j in range(len(cs))],
                                  sample="S%d" % i), "members")
        p["vcfs"].append(path)
    return p, contig


def restated_regions(p, trimming=True):
    calls = [fo.load_calls(v, "", trimming) for v in p["vcfs"]]
    return fo.generate_multi_regions(calls, fo.read_bed(p["bed"]), fo.read_fasta(p["fa"]))


def assert_same_multi(mb, regions):
    assert mb.n_regions == len(regions)
    k = mb.n_inputs
    got = mb.regions()
    for g, w in zip(got, regions):
        assert (g["region_id"], g["contig"], g["start"], g["end"]) == (w["region_id"], w["contig"], w["start"], w["end"])
        for i in range(k):
            assert [(v[0], v[1].decode(), v[2].decode(), v[3], v[4], v[5]) for v in g["inputs"][i]] == \
                   [(c["pos"], c["a0"], c["a1"], VT[c["type"]], ZYG[c["zyg"]], c["raw"]) for c in w["inputs"][i]]


def oracle_results(oracle, mb, contigs, config):
    """classification of every region: pairs from the C oracle, decision from the Python restatement"""
    regions = mb.regions()
    batch, owner = pair_batch(regions)
    st, ex = oracle_lib.optimize_pairs(oracle, batch, contigs, config.max_branch_factor, threads=8)
    pair = {o: (int(s), int(e)) for o, s, e in zip(owner, st, ex)}
    out = []
    k = mb.n_inputs
    for m, reg in enumerate(regions):
        if any(pair[(m, i, j)][0] != 0 for i in range(k) for j in range(i + 1, k)):
            out.append(None)
            continue
        out.append(mo.classify([len(v) for v in reg["inputs"]], lambda i, j: pair[(m, i, j)][1], config.no_conflict_enabled, config.majority_voting_enabled,
                               config.conflict_selection))
    return out


def as_result(results, k):
    st = np.array([0 if r is not None else 3 for r in results], np.int32)
    cls = np.array([CLS_CODE[r[0]] if r is not None else 0 for r in results], np.uint8)
    mem = np.zeros(len(results), np.uint64)
    for m, r in enumerate(results):
        if r is None:
            continue
        if r[0] in ("no_conflict", "majority"):
            mem[m] = sum(1 << i for i in r[1])
        elif r[0] == "conflict_select":
            mem[m] = r[1]
    return MergeResult(st, cls, mem, k)


def tbx_fetch(path, chrom, beg, end, fmt):
    """record lines of a BGZF text file overlapping [beg, end) of chrom, found THROUGH its .tbi; fmt 2 = VCF, 0x10000 = BED"""
    blocks = bgzf_blocks(path)
    by_off = {o: raw for o, raw in blocks}
    tbi = b"".join(raw for _, raw in bgzf_blocks(path + ".tbi"))
    assert tbi[:4] == b"TBI\x01"
    n_ref, f, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", tbi, 4)
    assert (f, col_seq, col_beg, col_end, meta, skip) == ((2, 1, 2, 0, ord("#"), 0) if fmt == 2 else (0x10000, 1, 2, 3, ord("#"), 0))
    names = tbi[36:36 + l_nm].split(b"\0")[:-1]
    assert len(names) == n_ref
    at = 36 + l_nm
    found = None
    for name in names:
        n_bin = struct.unpack_from("<i", tbi, at)[0]
        at += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", tbi, at)
            at += 8
            bins[b] = [struct.unpack_from("<QQ", tbi, at + 16 * c) for c in range(n_chunk)]
            at += 16 * n_chunk
        n_intv = struct.unpack_from("<i", tbi, at)[0]
        at += 4
        linear = list(struct.unpack_from("<%dQ" % n_intv, tbi, at))
        at += 8 * n_intv
        if name.decode() == chrom:
            found = (bins, linear)
    assert at == len(tbi)
    if found is None:
        return []
    bins, linear = found
    min_off = linear[min(beg >> 14, len(linear) - 1)] if linear else 0
    offs = sorted(by_off)
    recs = set()
    for b in reg2bins(beg, end):
        for cb, ce in bins.get(b, []):
            if ce <= min_off:
                continue
            kblk, pos, text = offs.index(cb >> 16), cb & 0xFFFF, b""
            while (offs[kblk] << 16 | pos) < ce:
                raw = by_off[offs[kblk]]
                last = offs[kblk] == ce >> 16
                text += raw[pos:(ce & 0xFFFF) if last else len(raw)]
                if last:
                    break
                kblk, pos = kblk + 1, 0
            for line in text.decode().splitlines():
                fld = line.split("\t")
                lo, hi = (int(fld[1]) - 1, int(fld[1]) - 1 + len(fld[3])) if fmt == 2 else (int(fld[1]), int(fld[2]))
                if fld[0] == chrom and lo < end and hi > beg:
                    recs.add(line)
    return sorted(recs, key=lambda l: (int(l.split("\t")[1]), l))


def check_outputs(out_dir, primary_vcf, regions, results, tags, sample, version=None, command=None):
    text = gzip.open(os.path.join(out_dir, "passing.vcf.gz"), "rt").read()
    lines = text.splitlines()
    meta = [l for l in lines if l.startswith("##")]
    src_meta = [l for l in gzip.open(primary_vcf, "rt").read().splitlines() if l.startswith("##")]
    # header layout of the reference's VCF library: file format, INFO, FILTER, FORMAT, ALT, contig groups, then the other lines
    group = lambda key: [l for l in src_meta if l.startswith("##%s=<" % key)]
    other = [l for l in src_meta if not l.startswith("##fileformat=") and not any(l.startswith("##%s=<" % k) for k in ("INFO", "FILTER", "FORMAT", "ALT", "contig"))]
    want = [l for l in src_meta if l.startswith("##fileformat=")] + group("INFO") + \
        ['##INFO=<ID=SOURCES,Number=.,Type=String,Description="List of tools or technologies that called the same record">',
         '##INFO=<ID=MR,Number=1,Type=String,Description="The reason this record was allowed in the merge">'] + group("FILTER") + group("FORMAT") + \
        ['##FORMAT=<ID=RI,Number=1,Type=Integer,Description="Region ID for the comparison">'] + group("ALT") + group("contig") + other
    assert meta[:len(want)] == want and len(meta) == len(want) + 2
    extra = meta[len(want):]
    assert extra[0].startswith('##aardvark_version="') and extra[1].startswith('##aardvark_command="')
    if version is not None:
        assert extra[0] == '##aardvark_version="%s"' % version and extra[1] == '##aardvark_command="%s"' % command
    assert lines[len(meta)] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample
    records = lines[len(meta) + 1:]
    assert records == mo.passing_vcf_records(regions, results, tags)
    passing, failed = mo.region_bed_lines(regions, results)
    got_pass = gzip.open(os.path.join(out_dir, "regions.bed.gz"), "rt").read().splitlines()
    got_fail = gzip.open(os.path.join(out_dir, "failed_regions.bed.gz"), "rt").read().splitlines()
    assert got_pass == passing and got_fail == failed
    key = lambda l: (int(l.split("\t")[1]), l)
    for chrom in sorted({r["chrom"] for r in regions}) + ["chrNone"]:
        vc = [r for r in records if r.split("\t")[0] == chrom]
        assert tbx_fetch(os.path.join(out_dir, "passing.vcf.gz"), chrom, 0, 1 << 29, 2) == sorted(set(vc), key=key)
        for name, want in (("regions.bed.gz", passing), ("failed_regions.bed.gz", failed)):
            mine = [l for l in want if l.split("\t")[0] == chrom]
            path = os.path.join(out_dir, name)
            assert tbx_fetch(path, chrom, 0, 1 << 29, 0x10000) == sorted(set(mine), key=key)
            for beg, end in ((0, 1000), (50_000, 50_001), (70_000, 130_000), (199_000, 200_000)):
                hit = [l for l in mine if int(l.split("\t")[1]) < end and int(l.split("\t")[2]) > beg]
                assert tbx_fetch(path, chrom, beg, end, 0x10000) == sorted(set(hit), key=key)
        for beg, end in ((0, 1000), (50_000, 50_400), (70_000, 130_000)):
            hit = [r for r in vc if int(r.split("\t")[1]) - 1 < end and int(r.split("\t")[1]) - 1 + len(r.split("\t")[3]) > beg]
            assert tbx_fetch(os.path.join(out_dir, "passing.vcf.gz"), chrom, beg, end, 2) == sorted(set(hit), key=key)
    return records, passing, failed


def test_merge_feed_matches_the_restatement(tmp_path):
    p, contig = write_case(tmp_path)
    genome = feeder.Genome(p["fa"])
    for trimming in (True, False):
        feed = feeder.feed_merge(p["vcfs"], p["bed"], genome, enable_trimming=trimming)
        regions, loaded = restated_regions(p, trimming)
        assert_same_multi(feed.batch, regions)
        assert list(feed.loaded) == loaded and len(regions) > 100
        # the feeder's own packed form (avf_feed_pack_multi) = the wide batch narrowed by the Python class
        from aardvark_amd.merge import PackedMultiBatch
        ref = PackedMultiBatch.from_multi(feed.batch)
        assert feed.packed is not None and (feed.packed.n_regions, feed.packed.n_variants, feed.packed.n_inputs) == (ref.n_regions, ref.n_variants, ref.n_inputs)
        for name in PackedMultiBatch.FIELDS:
            a, b = getattr(feed.packed, name), getattr(ref, name)
            assert (a is None and b is None) or np.array_equal(a, b), name
    # one input is the compare feed's truth side: the same windows when the second input is the same file
    one = feeder.feed_merge(p["vcfs"][:1], p["bed"], genome)
    assert one.batch.n_inputs == 1 and one.batch.n_regions > 0
    # samples by name, a different gap
    feed = feeder.feed_merge(p["vcfs"][:2], p["bed"], genome, samples=["S0", "S1"], min_variant_gap=7)
    regions, _ = fo.generate_multi_regions([fo.load_calls(p["vcfs"][0], "S0"), fo.load_calls(p["vcfs"][1], "S1")], fo.read_bed(p["bed"]), fo.read_fasta(p["fa"]), 7)
    assert_same_multi(feed.batch, regions)


def test_feed_in_two_halves(tmp_path):
    """avf_calls_load + avf_feed_from_calls (the tools load the call sets while the genome is still loading) give the feeds of the
    one-call forms, for compare and for merge; avf_vcf_sample_name names the samples"""
    from test_feeder import assert_same_batch
    p, contig = write_case(tmp_path, 200, 80_000, 3)
    genome = feeder.Genome(p["fa"])
    assert [feeder.vcf_sample_name(v) for v in p["vcfs"]] == ["S0", "S1", "S2"]
    with pytest.raises(feeder.FeederError, match="Sample index 1 does not exist"):
        feeder.vcf_sample_name(p["vcfs"][0], 1)
    calls = [feeder.Calls(v, sample="S%d" % i) for i, v in enumerate(p["vcfs"])]
    whole = feeder.feed_merge(p["vcfs"], p["bed"], genome)
    halves = feeder.feed_from_calls(calls, p["bed"], genome, merge=True)
    regions, loaded = restated_regions(p)
    assert_same_multi(halves.batch, regions)
    assert halves.loaded == whole.loaded == tuple(loaded) and np.array_equal(halves.var_record, whole.var_record)
    pair = feeder.feed_from_calls(calls[:2], p["bed"], genome, min_variant_gap=30)
    want = feeder.feed_compare(p["vcfs"][0], p["vcfs"][1], p["bed"], genome, min_variant_gap=30)
    assert_same_batch(pair.batch, want.batch)
    assert np.array_equal(pair.var_record, want.var_record) and np.array_equal(pair.var_alt_index, want.var_alt_index)
    with pytest.raises(feeder.FeederError, match="exactly two inputs"):
        feeder.feed_from_calls(calls, p["bed"], genome)
    with pytest.raises(feeder.FeederError, match="High confidence regions are currently required"):
        feeder.feed_from_calls(calls[:2], None, genome)
    with pytest.raises(feeder.FeederError, match="NOPE"):
        feeder.Calls(p["vcfs"][0], sample="NOPE")


def test_merge_feed_errors(tmp_path):
    p, _ = write_case(tmp_path, 30, 20_000, 2)
    genome = feeder.Genome(p["fa"])
    with pytest.raises(feeder.FeederError, match="Must provide at least 1 VCF"):
        feeder.feed_merge([], p["bed"], genome)
    with pytest.raises(feeder.FeederError, match="High confidence regions are currently required"):
        feeder.feed_merge(p["vcfs"], None, genome)
    with pytest.raises(feeder.FeederError, match="NOPE"):
        feeder.feed_merge(p["vcfs"], p["bed"], genome, samples=["S0", "NOPE"])
    with pytest.raises(feeder.FeederError, match="at most 64"):
        feeder.feed_merge(p["vcfs"][:1] * 65, p["bed"], genome)
    with pytest.raises(feeder.FeederError):
        feeder.feed_merge(p["vcfs"] + [str(tmp_path / "missing.vcf")], p["bed"], genome)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_merge_classification_outputs_and_summary(tmp_path, oracle, cfg):
    """host classification (avk_merge_classify on the oracle's pair results) against the restated decision; then the writers"""
    p, contig = write_case(tmp_path)
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_merge(p["vcfs"], p["bed"], genome)
    mb = feed.batch
    config = MergeConfig(**cfg)
    contigs = genome.contigs()
    want = oracle_results(oracle, mb, contigs, config)
    got = solve_merge_regions(lambda b, mbf: oracle_lib.optimize_pairs(oracle, b, contigs, mbf, threads=8), mb.regions(), config)
    assert [g[1] if g[0] == 0 else None for g in got] == want
    kinds = {w[0] for w in want if w}
    assert "identical" in kinds and len(kinds) >= 2
    regions, _ = restated_regions(p)
    # pretend one region failed in the solver: it is written nowhere
    want[3] = None
    res = as_result(want, mb.n_inputs)
    tags = ["hifi", "ont", "vcf_2", "with,comma"]
    feeder.write_merge_outputs(p["out"], p["vcfs"][0], genome, mb, res, tags=tags, version="v-test", command_line="merge --x")
    records, passing, failed = check_outputs(p["out"], p["vcfs"][0], regions, want, tags, "S0", "v-test", "merge --x")
    assert len(passing) + len(failed) == len(regions) - 1 and records
    assert not any(l.endswith("_%d" % regions[3]["region_id"]) for l in passing + failed)
    for ext, delim in (("tsv", "\t"), ("csv", ",")):
        path = str(tmp_path / ("merge_summary." + ext))
        feeder.write_merge_summary(path, mb, res, tags=tags)
        text = mo.merge_summary_text(regions, want, [t if delim == "\t" or "," not in t else '"%s"' % t for t in tags], lambda c: mo.TYPE_NAMES.index(c["type"]), delim)
        assert open(path).read() == text
    # a sample name for the output column; default tags
    feeder.write_merge_outputs(p["out"] + "/again/deeper", p["vcfs"][0], genome, mb, res, sample_name="OUT")
    check_outputs(p["out"] + "/again/deeper", p["vcfs"][0], regions, want, ["vcf_%d" % i for i in range(4)], "OUT")


def test_merge_summary_key_order_and_empty_outputs(tmp_path):
    """rows follow the derive(Ord) order of (MergeClassification, VariantType, vcf_index); an all-failed job leaves empty files"""
    regs = [{"start": 10 * m, "end": 10 * m + 9, "inputs": [[(10 * m + 1, "A", "C", "Snv", "HomozygousAlternate")], [(10 * m + 2, "AC", "A", "Deletion", "HomozygousAlternate")],
                                                            [(10 * m + 3, "A", "AT", "Insertion", "UnphasedHeterozygous")]]} for m in range(7)]
    mb = MultiBatch.from_regions(regs)
    results = [("identical",), ("majority", [0, 2]), ("majority", [0, 1]), ("no_conflict", [1]), ("conflict_select", 2), ("different",), ("no_conflict", [0, 1, 2])]
    res = as_result(results, 3)
    path = str(tmp_path / "s.tsv")
    feeder.write_merge_summary(path, mb, res)
    rows = [l.split("\t") for l in open(path).read().splitlines()]
    assert rows[0] == ["merge_reason", "variant_type", "vcf_index", "vcf_label", "pass_variants", "fail_variants"]
    assert [r[0] for r in rows[1:]] == ["different"] * 3 + ["no_conflict_0_1_2"] * 3 + ["no_conflict_1"] * 3 + ["majority_0_1"] * 3 + ["majority_0_2"] * 3 + \
        ["conflict_select_2"] * 3 + ["identical"] * 3
    assert [r[1:4] for r in rows[1:4]] == [["Snv", "0", "vcf_0"], ["Insertion", "2", "vcf_2"], ["Deletion", "1", "vcf_1"]]  # Snv < Insertion < Deletion before the input index
    by = {(r[0], r[2]): (r[4], r[5]) for r in rows[1:]}
    assert by[("majority_0_2", "1")] == ("0", "1") and by[("majority_0_2", "2")] == ("1", "0") and by[("conflict_select_2", "0")] == ("0", "1")
    text = mo.merge_summary_text([dict(r, chrom="c", region_id=m, inputs=[[dict(pos=v[0], a0=v[1], a1=v[2], type=v[3], zyg=v[4]) for v in i] for i in r["inputs"]])
                                  for m, r in enumerate(regs)], results, ["vcf_0", "vcf_1", "vcf_2"], lambda c: mo.TYPE_NAMES.index(c["type"]))
    assert open(path).read() == text
    # nothing solved: no rows at all (the csv writer only emits its header with the first row), empty but valid outputs
    none = MergeResult(np.full(7, 3, np.int32), np.zeros(7, np.uint8), np.zeros(7, np.uint64), 3)
    feeder.write_merge_summary(path, mb, none)
    assert open(path).read() == ""
    fa, vcf = str(tmp_path / "e.fa"), str(tmp_path / "e.vcf")
    write_text(fa, ">c\n" + "ACGT" * 30 + "\n")
    write_text(vcf, vcf_text("c", [(3, "A", "C", "HomozygousAlternate")], sample="only"))
    genome = feeder.Genome(fa)
    out = str(tmp_path / "empty")
    feeder.write_merge_outputs(out, vcf, genome, mb, none)
    for name in ("passing.vcf.gz", "regions.bed.gz", "failed_regions.bed.gz"):
        body = [l for l in gzip.open(os.path.join(out, name), "rt").read().splitlines() if not l.startswith("#")]
        assert body == [] and os.path.exists(os.path.join(out, name + ".tbi"))
    with pytest.raises(feeder.FeederError, match="names input"):
        feeder.write_merge_summary(path, mb, as_result([("conflict_select", 5)] * 7, 3))


def merge_cli():
    return os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_merge")


def test_merge_tool_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p, _ = write_case(tmp_path, 30, 20_000, 2)
    r = subprocess.run([merge_cli(), "-r", p["fa"], "-i", p["vcfs"][0], "-i", p["vcfs"][1], "-b", p["bed"], "-o", p["out"]], capture_output=True, text=True)
    assert r.returncode == 70 and "cannot create the GPU context" in r.stderr
    assert not os.path.exists(os.path.join(p["out"], "passing.vcf.gz"))
    r = subprocess.run([merge_cli(), "-r", p["fa"], "-i", p["vcfs"][0], "-b", p["bed"], "-o", p["out"], "--conflict-select", "1"], capture_output=True, text=True)
    assert r.returncode == 78 and "--conflict-selection index is greater than number of provided VCFs" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CONFIGS)
def test_merge_batch_on_gpu_matches_oracle(tmp_path, oracle, cfg):
    import torch
    torch.cuda.init()
    import aardvark_amd
    from aardvark_amd.merge import merge_multi_batch
    p, contig = write_case(tmp_path, 1500, 600_000)
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_merge(p["vcfs"], p["bed"], genome)
    ctx = aardvark_amd.Context(0)
    ctx.upload_reference(genome.contigs())
    config = MergeConfig(**cfg)
    res = merge_multi_batch(ctx, feed.batch, config)
    want = oracle_results(oracle, feed.batch, genome.contigs(), config)
    assert [g[1] if g[0] == 0 else None for g in res.decoded()] == want


@pytest.mark.gpu
def test_merge_tool_end_to_end(tmp_path, oracle):
    """FASTA + BED + 4 VCFs -> aardvark_amd_merge on the GPU -> passing.vcf.gz, BED files, summary, cli_settings.json"""
    p, contig = write_case(tmp_path, 1500, 600_000)
    summary, debug = str(tmp_path / "summary.csv"), str(tmp_path / "debug")
    cmd = [merge_cli(), "-r", p["fa"]] + [x for v in p["vcfs"] for x in ("-i", v)] + ["-b", p["bed"], "-o", p["out"], "-t", "hifi", "-t", "ont", "--output-summary", summary,
                                                                                       "--output-debug", debug, "--merge-strategy", "all", "--conflict-select", "1",
                                                                                       "--batch-regions", "300"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    genome = feeder.Genome(p["fa"])
    regions, loaded = restated_regions(p)
    feed = feeder.feed_merge(p["vcfs"], p["bed"], genome)
    config = MergeConfig(no_conflict_enabled=True, majority_voting_enabled=True, conflict_selection=1)
    want = oracle_results(oracle, feed.batch, genome.contigs(), config)
    tags = ["hifi", "ont", "vcf_2", "vcf_3"]
    check_outputs(p["out"], p["vcfs"][0], regions, want, tags, "S0")
    assert open(summary).read() == mo.merge_summary_text(regions, want, tags, lambda c: mo.TYPE_NAMES.index(c["type"]), ",")
    assert "Solved:error blocks: %d : 0" % len(regions) in r.stderr
    for i, n in enumerate(loaded):
        assert "Loaded %d variants from input #%d." % (n, i) in r.stderr
    js = json.load(open(os.path.join(debug, "cli_settings.json")))
    assert js["vcf_filenames"] == p["vcfs"] and js["vcf_tags"] == tags and js["merge_strategy"] == "AllOptions" and js["enable_voting"] is True
    assert js["conflict_selection"] == 1 and js["take_blocks"] == 2 ** 64 - 1 and js["min_variant_gap"] == 50
    # the exact strategy: fewer passing regions, the rest in failed_regions.bed.gz; --skip / --take leave the other regions out
    r = subprocess.run([merge_cli(), "-r", p["fa"]] + [x for v in p["vcfs"] for x in ("-i", v)] + ["-b", p["bed"], "-o", p["out"] + "2", "--skip", "5", "--take", "40"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want2 = oracle_results(oracle, feed.batch, genome.contigs(), MergeConfig())
    want2 = [w if 5 <= m < 45 else None for m, w in enumerate(want2)]
    check_outputs(p["out"] + "2", p["vcfs"][0], regions, want2, ["vcf_%d" % i for i in range(4)], "S0")
    # the tool hands the packed form to avk_merge_packed by default (here in batches of 300 regions: avf_packed_multi_slice); --batch-form wide gives the same files
    r = subprocess.run(cmd[:cmd.index("-o")] + ["-o", p["out"] + "3"] + cmd[cmd.index("-o") + 2:] + ["--batch-form", "wide"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for name in ("passing.vcf.gz", "regions.bed.gz", "failed_regions.bed.gz"):
        a, b = (gzip.open(os.path.join(d, name), "rb").read() for d in (p["out"], p["out"] + "3"))
        strip = lambda x: b"\n".join(l for l in x.split(b"\n") if not l.startswith(b"##aardvark_command"))
        assert strip(a) == strip(b), name
