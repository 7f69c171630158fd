"""The C++ feeder (FASTA / BED / VCF -> region batch) and summary writer of libaardvark_feeder.so against the plain-Python
restatement of the reference (oracle/feeder_oracle.py), the synthetic generator's clustering, and — pinned by the
reference's own known answers — the 8 solve_compare_region test regions sent through VCF files."""
import gzip
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import feeder_oracle as fo  # noqa: E402
import oracle_lib  # noqa: E402
import scenarios  # noqa: E402
from aardvark_amd import feeder, synth  # noqa: E402
from aardvark_amd._abi import F, VT, ZYG, RegionBatch  # noqa: E402
from test_oracle_golden import check_region_expectations  # noqa: E402

GT_OF = {"HomozygousAlternate": "1/1", "UnphasedHeterozygous": "0/1", "PhasedHet01": "0|1", "PhasedHet10": "1|0"}
ZNAME = {v: k for k, v in ZYG.items()}
HEADER = "##fileformat=VCFv4.2\n##contig=<ID={c}>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t{s}\n"


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def write_text(path, text, mode="plain"):
    data = text.encode()
    if mode == "plain":
        open(path, "wb").write(data)
    elif mode == "gz":
        open(path, "wb").write(gzip.compress(data))
    else:  # several gzip members back to back, like BGZF blocks
        cut = [0, len(data) // 3, 2 * len(data) // 3, len(data)]
        open(path, "wb").write(b"".join(gzip.compress(data[a:b]) for a, b in zip(cut[:-1], cut[1:])))


def bgzf_bytes(data, block=0xff00, level=6):
    """BGZF as bgzip writes it: independent gzip members of at most 64 KiB with the BC extra field, then the empty end-of-file block"""
    import struct
    import zlib
    out = []
    for at in list(range(0, len(data), block)) + [None]:
        chunk = b"" if at is None else data[at:at + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(payload) + 25) + payload +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def vcf_text(chrom, calls, sample="S1"):
    """calls: [(pos0, ref, alt, zygosity name)]"""
    lines = [HEADER.format(c=chrom, s=sample)]
    for pos, ref, alt, zyg in calls:
        lines.append("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (chrom, pos + 1, ref, alt, GT_OF[zyg]))
    return "".join(lines)


def batch_of(regions):
    """feeder_oracle regions -> RegionBatch"""
    conv = lambda c: (c["pos"], c["a0"], c["a1"], c["type"], c["zyg"], c["raw"])
    return RegionBatch.from_regions([dict(region_id=r["region_id"], contig=r["contig"], start=r["start"], end=r["end"],
                                          truth=[conv(c) for c in r["truth"]], query=[conv(c) for c in r["query"]]) for r in regions])


def assert_same_batch(a, b):
    assert a.n_regions == b.n_regions and a.n_variants == b.n_variants
    for f in ("region_id", "contig_idx", "start", "end", "t_cnt", "q_cnt", "t_off", "q_off", "var_pos", "var_type", "var_zyg", "var_raw_space", "a0_len", "a1_len"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for v in range(a.n_variants):
        for off, ln in (("a0_off", "a0_len"), ("a1_off", "a1_len")):
            sa = a.allele_bytes[int(getattr(a, off)[v]):int(getattr(a, off)[v]) + int(getattr(a, ln)[v])]
            sb = b.allele_bytes[int(getattr(b, off)[v]):int(getattr(b, off)[v]) + int(getattr(b, ln)[v])]
            assert bytes(sa) == bytes(sb)


def test_library_exports_every_declared_symbol():
    lib = feeder.load_library()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "aardvark_feeder.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(avf_[a-z0-9_]+)\s*\(", src)))
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


EDGE_FASTA = ">chrA first contig\n" + "ACGTTGCAAC" * 40 + "\n>chrB\n" + "GATTACA" * 30 + "\n>chrUnused\nACGT\n"
EDGE_BED = "chrB\t5\t150\nchrA\t200\t390\nchrA\t10\t120\nchrA\t130\t180\n#comment\nchrB\t160\t205\n"  # chrB first, chrA unsorted
EDGE_VCF_T = HEADER.format(c="chrA", s="OTHER\tS1") + "".join([
    "chrA\t5\t.\tT\tG\t.\t.\t.\tGT\t0/1\t1/1\n",                      # before the first interval: dropped
    "chrA\t12\t.\tT\tG,C\t.\t.\t.\tGT:DP\t0/0:3\t1|2:9\n",            # multi-allelic, phased: two calls
    "chrA\t15\t.\tTGC\tT\t.\t.\t.\tGT\t./.\t1\n",                     # hemizygous -> homozygous
    "chrA\t40\t.\tCA\tCAA,*\t.\t.\t.\tGT\t0/1\t2/1\n",                 # '*' allele skipped, insertion kept; no suffix left after trimming? CA/CAA -> C/CA
    "chrA\t60\t.\tA\t<DEL>\t.\t.\tSVTYPE=DEL\tGT\t0/1\t0/1\n",         # symbolic: skipped
    "chrA\t70\t.\tAC\tGC\t.\t.\t.\tGT\t0/1\t0|1\n",                    # trims to A>G (SNV), raw space 2
    "chrA\t80\t.\tG\tT\t.\t.\t.\tGT\t0/1\t.\n",                       # missing GT: no-op
    "chrA\t90\t.\tG\tT\t.\t.\t.\tGT\t0/1\t0/0\n",                      # hom ref: nothing
    "chrA\t118\t.\tGCAACA\tG\t.\t.\t.\tGT\t0/1\t0/1\n",                # runs past the interval end (120): overlapping, dropped
    "chrA\t125\t.\tG\tT\t.\t.\t.\tGT\t0/1\t1/1\n",                     # between intervals: dropped
    "chrA\t135\t.\tACGTTGCAACACG\tA\t.\t.\tSVTYPE=DEL\tGT\t0/1\t1/0\n",  # sequence-resolved SV deletion
    "chrA\t150\t.\tC\tCGGGG\t.\t.\tTRID=tr1\tGT\t0/1\t0/1\n",          # tandem-repeat expansion
    "chrA\t160\t.\tCAAC\tC\t.\t.\tTRID=tr2\tGT\t0/1\t1|0\n",           # tandem-repeat contraction
    "chrA\t170\t.\tA\tAT\t.\t.\tSVTYPE=DUP\tGT\t0/1\t0/1\n",           # unsupported SV kind: skipped
    "chrA\t210\t.\tC\tA\t.\t.\t.\tGT\t0/1\t0/1\n",
    "chrA\t262\t.\tC\tA\t.\t.\t.\tGT\t0/1\t0/1\n",                     # 52 bp later: its own window... (210+1+50 = 260 <= 261)
    "chrA\t300\t.\tC\tCTT\t.\t.\t.\tGT\t0/1\t1/1\n",
    "chrA\t395\t.\tC\tA\t.\t.\t.\tGT\t0/1\t1/1\n",                     # after the last interval
    "chrB\t3\t.\tT\tA\t.\t.\t.\tGT\t0/1\t1/1\n",                       # before chrB's span
    "chrB\t10\t.\tA\tC\t.\t.\t.\tGT\t0/1\t1/1\n",
    "chrB\t200\t.\tC\tA\t.\t.\t.\tGT\t0/1\t0/1\n",
    "chrC\t10\t.\tA\tC\t.\t.\t.\tGT\t0/1\t1/1\n",                      # chromosome without BED intervals
])
EDGE_VCF_Q = HEADER.format(c="chrA", s="Q") + "".join([
    "chrA\t12\t.\tT\tG\t.\t.\t.\tGT\t0/1\n",
    "chrA\t12\t.\tT\tC\t.\t.\t.\tGT\t1/0\n",
    "chrA\t16\t.\tGC\tG\t.\t.\t.\tGT\t1/1\n",
    "chrA\t41\t.\tA\tAA\t.\t.\t.\tGT\t0/1\n",
    "chrA\t70\t.\tA\tG\t.\t.\t.\tGT\t1|0\n",
    "chrA\t300\t.\tC\tCT\t.\t.\t.\tGT\t1/1\n",
    "chrB\t10\t.\tA\tC\t.\t.\t.\tGT\t0/1\n",
    "chrB\t204\t.\tAC\tA\t.\t.\t.\tGT\t0/1\n",                          # ends exactly at the span end (205): contained
])


@pytest.mark.parametrize("mode", ["plain", "gz", "members"])
def test_edge_case_files_match_the_restatement(tmp_path, mode):
    paths = {}
    for name, text in (("ref.fa", EDGE_FASTA), ("hc.bed", EDGE_BED), ("truth.vcf", EDGE_VCF_T), ("query.vcf", EDGE_VCF_Q)):
        paths[name] = str(tmp_path / (name + ("" if mode == "plain" else ".gz")))
        write_text(paths[name], text, mode)
    genome = feeder.Genome(paths["ref.fa"])
    contigs = fo.read_fasta(paths["ref.fa"])
    assert genome.names == [n for n, _ in contigs] == ["chrA", "chrB", "chrUnused"]
    assert [bytes(c).decode() for c in genome.contigs()] == [s for _, s in contigs]
    for trim in (True, False):
        feed = feeder.feed_compare(paths["truth.vcf"], paths["query.vcf"], paths["hc.bed"], genome, truth_sample="S1", enable_trimming=trim)
        regions, loaded = fo.generate_regions(fo.load_calls(paths["truth.vcf"], "S1", trim), fo.load_calls(paths["query.vcf"], "", trim),
                                              fo.read_bed(paths["hc.bed"]), contigs)
        assert_same_batch(feed.batch, batch_of(regions))
        assert list(feed.loaded) == loaded
        flat = [c for r in regions for c in r["truth"] + r["query"]]
        assert [int(x) for x in feed.var_record] == [c["record"] for c in flat]
        assert [int(x) for x in feed.var_alt_index] == [c["alt_index"] for c in flat]
        if trim:  # spot checks of the rules themselves, independent of the restatement
            b = feed.batch
            assert [int(x) for x in b.contig_idx[:2]] == [1, 1]          # chrB comes first in the BED
            names = {(int(b.contig_idx[r]), int(b.start[r]), int(b.end[r])) for r in range(b.n_regions)}
            assert (0, 159, 260) in names and (0, 211, 350) in names     # 210 and 262 (1-based) do not share a window; 262 and 300 do
            t_types = [int(x) for x in b.var_type[:]]
            assert VT["SvDeletion"] in t_types and VT["TrExpansion"] in t_types and VT["TrContraction"] in t_types
            i = list(b.var_pos).index(69)                                # AC>GC trimmed to A>G, raw allele space kept
            assert int(b.a0_len[i]) == 1 and int(b.var_raw_space[i]) == 2 and int(b.var_zyg[i]) == ZYG["PhasedHet01"]


def test_mapped_fasta_loader_matches_the_line_reader(tmp_path, monkeypatch):
    """plain FASTA files are parsed from a mapping by several threads, gzip ones line by line: same contigs either way, with CRLF
    line ends, empty lines, '>' inside header text, no final newline, and piece boundaries at every possible offset"""
    rng = np.random.default_rng(5)
    def seq(n):
        return "".join(rng.choice(list("ACGTNacgt"), size=n))
    texts = [">c1 desc > more\n" + seq(70) + "\n" + seq(70) + "\n\n" + seq(13) + "\n>c2\tx\n" + seq(5) + "\n>empty\n>c3\n" + seq(200),
             "\n\r\n>a\r\n" + seq(60) + "\r\n" + seq(60) + "\r\n\r\n" + seq(7) + "\r\n>b x\r\n" + seq(31) + "\r",
             ">only\n" + seq(1000) + "\n"]
    for ti, text in enumerate(texts):
        plain, gz = str(tmp_path / ("f%d.fa" % ti)), str(tmp_path / ("f%d.fa.gz" % ti))
        write_text(plain, text)
        write_text(gz, text, "gz")
        want = feeder.Genome(gz, case="raw")
        ref = dict(fo.read_fasta(gz))
        for piece in ("1", "2", "3", "7", "64", "1000000"):
            monkeypatch.setenv("AVF_FASTA_PIECE", piece)
            got = feeder.Genome(plain, case="raw")
            assert got.names == want.names
            for a, b, name in zip(got.contigs(), want.contigs(), got.names):
                assert bytes(a) == bytes(b) == ref[name].encode()
        # bgzip-compressed FASTA: blocks inflated by several threads, then the same parser
        for block in (50, 4096):
            bg = str(tmp_path / ("f%d_%d.fa.bgz" % (ti, block)))
            open(bg, "wb").write(bgzf_bytes(text.encode(), block))
            monkeypatch.setenv("AVF_FASTA_PIECE", "33")
            got = feeder.Genome(bg, case="raw")
            assert got.names == want.names and all(bytes(a) == bytes(b) for a, b in zip(got.contigs(), want.contigs()))
            up = feeder.Genome(bg)  # the default folds soft-masked bases to upper case, whatever the loader
            assert all(bytes(a) == bytes(b).upper() for a, b in zip(up.contigs(), want.contigs()))
        for path in (plain, gz):
            assert all(bytes(a) == bytes(b).upper() for a, b in zip(feeder.Genome(path).contigs(), want.contigs()))
    damaged = bytearray(bgzf_bytes(texts[2].encode(), 200))
    damaged[len(damaged) // 2] ^= 0x11
    dpath = str(tmp_path / "damaged.fa.gz")
    open(dpath, "wb").write(bytes(damaged))
    with pytest.raises(feeder.FeederError):
        feeder.Genome(dpath)
    bad = str(tmp_path / "bad.fa")
    write_text(bad, "\nACGT\n>c\nAC\n")
    with pytest.raises(feeder.FeederError, match="sequence before the first header"):
        feeder.Genome(bad)


def test_soft_masked_reference_is_compared_as_upper_case(tmp_path, oracle):
    """GRCh38-style soft masking: the same calls against an upper-case contig and against its soft-masked twin give the same results with
    the default --reference-case upper; with raw bytes an ALT that re-states masked reference bases is charged for the case difference"""
    rng = np.random.default_rng(11)
    contig = "".join(rng.choice(list("ACGT"), size=600))
    masked = contig[:200] + contig[200:420].lower() + contig[420:]
    # MNV at 0-based 300 (inside the masked stretch): REF = 3 reference bases, ALT keeps the first two (upper case in the VCF) and changes the third
    ref3 = contig[300:303]
    alt3 = ref3[:2] + ("A" if ref3[2] != "A" else "C")
    calls = [(300, ref3, alt3, "HomozygousAlternate"), (449, contig[449], "A" if contig[449] != "A" else "C", "UnphasedHeterozygous")]  # 0-based
    write_text(str(tmp_path / "truth.vcf"), vcf_text("c", calls))
    write_text(str(tmp_path / "query.vcf"), vcf_text("c", calls))
    write_text(str(tmp_path / "r.bed"), "c\t0\t600\n")
    res = {}
    for name, seq, case in (("upper_file", contig, "upper"), ("masked_default", masked, "upper"), ("masked_raw", masked, "raw")):
        write_text(str(tmp_path / (name + ".fa")), ">c\n" + seq + "\n")
        g = feeder.Genome(str(tmp_path / (name + ".fa")), case=case)
        feed = feeder.feed_compare(str(tmp_path / "truth.vcf"), str(tmp_path / "query.vcf"), str(tmp_path / "r.bed"), g, enable_trimming=False)
        res[name] = oracle_lib.compare_batch(oracle, feed.batch, g.contigs())
    assert res["masked_default"].diff(res["upper_file"]) == []
    bp = lambda r: int(r.tally[F["BP_TRUTH_TP"]])
    assert bp(res["masked_raw"]) > bp(res["upper_file"])  # raw bytes: "ac" vs "AC" counts as two more edits on each haplotype


def test_block_parallel_vcf_reader_matches_the_sequential_one(tmp_path, monkeypatch):
    """VCFs are decompressed on one thread and parsed in blocks of whole lines by others: same calls, same record numbers, same errors
    as the line-by-line reader, whatever the block size"""
    paths = {}
    for name, text in (("ref.fa", EDGE_FASTA), ("hc.bed", EDGE_BED), ("truth.vcf", EDGE_VCF_T), ("query.vcf", EDGE_VCF_Q)):
        paths[name] = str(tmp_path / name)
        write_text(paths[name], text, "members" if name.endswith("vcf") else "plain")
    p, contig, want_batch = write_case_files(tmp_path, 800, 400_000)
    cases = [(paths["ref.fa"], paths["truth.vcf"], paths["query.vcf"], paths["hc.bed"], "S1", True), (p["fa"], p["t"], p["q"], p["bed"], "", False)]
    for fa, t, q, bed, sample, trimming in cases:
        genome = feeder.Genome(fa)
        monkeypatch.setenv("AVF_SEQUENTIAL_VCF", "1")
        want = feeder.feed_compare(t, q, bed, genome, truth_sample=sample, enable_trimming=trimming)
        monkeypatch.delenv("AVF_SEQUENTIAL_VCF")
        for block in ("16", "100", "5000", "4194304"):
            monkeypatch.setenv("AVF_VCF_BLOCK", block)
            got = feeder.feed_compare(t, q, bed, genome, truth_sample=sample, enable_trimming=trimming)
            assert_same_batch(got.batch, want.batch)
            assert np.array_equal(got.var_record, want.var_record) and np.array_equal(got.var_alt_index, want.var_alt_index) and got.loaded == want.loaded
    # an error deep in the file is reported exactly like the sequential reader reports it
    genome = feeder.Genome(p["fa"])
    lines = gzip.open(p["t"], "rt").read().splitlines()
    lines[len(lines) // 2] = lines[len(lines) // 2].replace("\tGT\t", "\tDP\t")
    bad = str(tmp_path / "bad_mid.vcf")
    write_text(bad, "\n".join(lines) + "\n")
    msgs = []
    for env in ({"AVF_SEQUENTIAL_VCF": "1"}, {"AVF_VCF_BLOCK": "200"}, {}):
        for k in ("AVF_SEQUENTIAL_VCF", "AVF_VCF_BLOCK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pytest.raises(feeder.FeederError) as e:
            feeder.feed_compare(bad, p["q"], p["bed"], genome)
        msgs.append(str(e.value))
    assert "Missing GT" in msgs[0] and msgs[0] == msgs[1] == msgs[2]


def test_bgzf_vcfs_are_inflated_and_parsed_in_groups(tmp_path, monkeypatch, capfd):
    """BGZF inputs: groups of blocks are inflated and parsed by the worker threads, the lines cut by group boundaries are put together
    afterwards — same calls and record numbers as the line-by-line reader for every block and group size, CRLF and a missing final
    line feed included; a damaged block or a plain gzip file goes the ordinary way"""
    p, contig, want_batch = write_case_files(tmp_path, 600, 300_000)
    genome = feeder.Genome(p["fa"])
    texts = {k: gzip.open(p[k], "rb").read() for k in ("t", "q")}
    monkeypatch.setenv("AVF_SEQUENTIAL_VCF", "1")
    want = feeder.feed_compare(p["t"], p["q"], p["bed"], genome, enable_trimming=False)
    monkeypatch.delenv("AVF_SEQUENTIAL_VCF")
    variants = [("lf", lambda b: b), ("crlf", lambda b: b.replace(b"\n", b"\r\n")), ("no_final_lf", lambda b: b.rstrip(b"\n"))]
    for name, change in variants:
        for block in (97, 1000, 0xff00):
            paths = {}
            for k in ("t", "q"):
                paths[k] = str(tmp_path / ("%s_%s_%d.vcf.gz" % (k, name, block)))
                open(paths[k], "wb").write(bgzf_bytes(change(texts[k]), block))
            assert gzip.open(paths["t"], "rb").read() == change(texts["t"])  # an ordinary gzip reader sees the text
            for group in ("1", "2", "7", "64"):
                monkeypatch.setenv("AVF_VCF_GROUP", group)
                monkeypatch.setenv("AVF_TIMING", "1")
                capfd.readouterr()
                got = feeder.feed_compare(paths["t"], paths["q"], p["bed"], genome, enable_trimming=False)
                monkeypatch.delenv("AVF_TIMING")
                assert capfd.readouterr().err.count("BGZF blocks in") == 2  # both files went through the group reader
                assert_same_batch(got.batch, want.batch)
                assert np.array_equal(got.var_record, want.var_record) and got.loaded == want.loaded
    # a damaged payload: the group reader steps aside and the ordinary reader reports the read error
    raw = bytearray(bgzf_bytes(texts["t"], 1000))
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "damaged.vcf.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(feeder.FeederError):
        feeder.feed_compare(bad, p["q"], p["bed"], genome)
    # an error inside a record is worded by the ordinary reader
    lines = texts["t"].decode().splitlines()
    lines[len(lines) // 2] = lines[len(lines) // 2].replace("\tGT\t", "\tDP\t")
    broken = str(tmp_path / "broken.vcf.gz")
    open(broken, "wb").write(bgzf_bytes(("\n".join(lines) + "\n").encode(), 500))
    monkeypatch.setenv("AVF_VCF_GROUP", "3")
    with pytest.raises(feeder.FeederError, match="Missing GT"):
        feeder.feed_compare(broken, p["q"], p["bed"], genome)
    with pytest.raises(feeder.FeederError, match="NOPE"):
        feeder.feed_compare(paths["t"], paths["q"], p["bed"], genome, truth_sample="NOPE")


def test_errors_are_reported(tmp_path):
    fa, bed = str(tmp_path / "r.fa"), str(tmp_path / "r.bed")
    write_text(fa, EDGE_FASTA)
    write_text(bed, "chrZ\t1\t50\n")
    t = str(tmp_path / "t.vcf")
    write_text(t, vcf_text("chrA", [(11, "T", "G", "HomozygousAlternate")]))
    g = feeder.Genome(fa)
    with pytest.raises(feeder.FeederError, match="not found in reference genome"):
        feeder.feed_compare(t, t, bed, g)
    with pytest.raises(feeder.FeederError, match="was not found in"):
        feeder.feed_compare(t, t, bed, g, truth_sample="nobody")
    with pytest.raises(feeder.FeederError, match="required"):
        feeder.feed_compare(t, t, "", g)
    bad = str(tmp_path / "bad.vcf")
    write_text(bad, HEADER.format(c="chrA", s="S1") + "chrA\t12\t.\tT\tG\t.\t.\t.\tGT\t0/1/1\n")
    write_text(bed, "chrA\t1\t50\n")
    with pytest.raises(feeder.FeederError, match="allele.len"):
        feeder.feed_compare(bad, t, bed, g)


def test_synthetic_call_sets_round_trip_through_vcf(tmp_path):
    """the generator's own clustering (a numpy mirror of RegionIterator) and the feeder agree on a chr20-like call set"""
    length = 400_000
    contig = synth.make_contig(length, 5)
    rng = np.random.default_rng(6)
    bed = synth.make_bed(length, 12, 0.9, rng)
    truth = synth.indel_truth(contig, bed, 1500, 7)
    query = synth.perturb_query(contig, bed, truth, 8, 20)
    want = synth.cluster_regions(length, bed, truth, query, 50)
    fa, bd, tv, qv = (str(tmp_path / n) for n in ("c.fa.gz", "c.bed", "t.vcf.gz", "q.vcf"))
    seq = bytes(contig).decode()
    write_text(fa, ">chr20 synthetic\n" + "\n".join(seq[i:i + 60] for i in range(0, length, 60)) + "\n", "members")
    write_text(bd, "".join("chr20\t%d\t%d\n" % (a, b) for a, b in bed))
    for path, cs, mode in ((tv, truth, "gz"), (qv, query, "plain")):
        write_text(path, vcf_text("chr20", [(int(cs.pos[i]), cs.ref[i].decode(), cs.alt[i].decode(), ZNAME[int(cs.zyg[i])]) for i in range(len(cs))]), mode)
    genome = feeder.Genome(fa)
    feed = feeder.feed_compare(tv, qv, bd, genome, enable_trimming=False)
    assert_same_batch(feed.batch, want)
    assert np.array_equal(genome.contigs()[0], contig)
    # the feeder's own packed form (avf_feed_pack) = the wide batch narrowed by the Python classes
    from aardvark_amd._abi import AvkPackedBatch, CompactBatch, PackedBatch
    ref = PackedBatch.from_compact(CompactBatch.from_region_batch(feed.batch))
    assert feed.packed is not None and feed.packed.n_regions == ref.n_regions and feed.packed.n_variants == ref.n_variants
    for name in PackedBatch.FIELDS:
        a, b = getattr(feed.packed, name), getattr(ref, name)
        assert (a is None and b is None) or np.array_equal(a, b), name
    assert feed.packed.var_raw_space is None  # untrimmed calls: the raw space is the longer allele
    # parts of it (avf_packed_slice): the arrays of regions [first, first + n) and of their calls, inside the whole
    import ctypes as C
    lib = feeder.load_library()
    h = C.c_void_p()
    assert lib.avf_feed_compare(os.fsencode(tv), b"", os.fsencode(qv), b"", os.fsencode(bd), genome.handle, 50, 0, C.byref(h)) == 0
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    alloc = feeder._ALLOC(lambda _u, nbytes: libc.malloc(nbytes))
    whole, part, v_first = AvkPackedBatch(), AvkPackedBatch(), C.c_uint64()
    lib.avf_feed_pack.argtypes = [C.c_void_p, feeder._ALLOC, C.c_void_p, C.POINTER(AvkPackedBatch)]
    lib.avf_packed_slice.argtypes = [C.c_void_p, C.POINTER(AvkPackedBatch), C.c_uint64, C.c_uint64, C.POINTER(AvkPackedBatch), C.POINTER(C.c_uint64)]
    assert lib.avf_feed_pack(h, alloc, None, C.byref(whole)) == 0
    addr = lambda ptr: C.cast(ptr, C.c_void_p).value
    b, n = feed.batch, feed.batch.n_regions
    for first, cnt in ((0, n), (0, 0), (n, 0), (n // 3, n // 2), (n - 1, 1), (5, 1)):
        assert lib.avf_packed_slice(h, C.byref(whole), first, cnt, C.byref(part), C.byref(v_first)) == 0
        v0 = int(b.t_off[first]) if first < n else b.n_variants
        v1 = int(b.t_off[first + cnt]) if first + cnt < n else b.n_variants
        a0 = int(b.a0_off[v0]) if v0 < b.n_variants else int(whole.allele_bytes_len)
        a1 = int(b.a0_off[v1]) if v1 < b.n_variants else int(whole.allele_bytes_len)
        assert (part.n_regions, part.n_variants, part.allele_bytes_len, v_first.value) == (cnt, v1 - v0, a1 - a0, v0)
        assert addr(part.start) - addr(whole.start) == 4 * first and addr(part.len) - addr(whole.len) == 2 * first and addr(part.contig_idx) - addr(whole.contig_idx) == 2 * first
        assert addr(part.t_cnt) - addr(whole.t_cnt) == first and addr(part.q_cnt) - addr(whole.q_cnt) == first
        assert addr(part.var_rel_pos) - addr(whole.var_rel_pos) == 2 * v0 and addr(part.var_type_zyg) - addr(whole.var_type_zyg) == v0
        assert addr(part.a0_len) - addr(whole.a0_len) == v0 and addr(part.a1_len) - addr(whole.a1_len) == v0 and addr(part.allele_bytes) - addr(whole.allele_bytes) == a0
        assert addr(part.var_raw_space) is None
    assert lib.avf_packed_slice(h, C.byref(whole), n, 1, C.byref(part), C.byref(v_first)) < 0
    lib.avf_feed_free(h)


def test_call_sets_outside_the_packed_form_are_reported(tmp_path):
    """an allele of more than 255 bases, or a window of 65,536 bases or more: avf_feed_pack returns 1 and the wide batch is what there is; trimmed calls keep
    their raw space in the packed form"""
    length = 200_000
    contig = synth.make_contig(length, 11)
    seq = bytes(contig).decode()
    fa, bd = str(tmp_path / "c.fa"), str(tmp_path / "c.bed")
    write_text(fa, ">chr20\n" + "\n".join(seq[i:i + 60] for i in range(0, length, 60)) + "\n")
    write_text(bd, "chr20\t0\t%d\n" % length)
    genome = feeder.Genome(fa)
    snv = lambda pos: (pos, seq[pos], "A" if seq[pos] != "A" else "C", "UnphasedHeterozygous")
    long_del = (5000, seq[5000:5000 + 300], seq[5000], "HomozygousAlternate")
    padded = (9000, seq[9000:9003], ("A" if seq[9000] != "A" else "C") + seq[9001:9003], "HomozygousAlternate")  # trimmed to one base, raw space 3
    cases = {"fits": ([snv(1000), padded], True), "long_allele": ([snv(1000), long_del], False)}
    for name, (calls, fits) in cases.items():
        tv = str(tmp_path / (name + ".vcf"))
        write_text(tv, vcf_text("chr20", calls))
        feed = feeder.feed_compare(tv, tv, bd, genome)
        assert (feed.packed is not None) == fits, name
        if fits:
            assert feed.packed.var_raw_space is not None and np.array_equal(feed.packed.var_raw_space, feed.batch.var_raw_space)
    tv = str(tmp_path / "fits.vcf")
    feed = feeder.feed_compare(tv, tv, bd, genome, min_variant_gap=60_000)  # one window of 69,003 bases
    assert feed.packed is None and feed.batch.n_regions == 1


def test_reference_known_answers_through_vcf_files(tmp_path, oracle):
    """the 8 solve_compare_region test regions as truth/query VCFs + BED on mock_chr1; the metrics the
    reference asserts come out of feeder -> solver (the window is the generated one, so sequences are not compared)."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "waffle_solver.json")))
    fa, bd = str(tmp_path / "mock.fa"), str(tmp_path / "mock.bed")
    write_text(fa, ">mock_chr1\n" + g["contig"] + "\n")
    write_text(bd, "mock_chr1\t0\t%d\n" % len(g["contig"]))
    genome = feeder.Genome(fa)
    tallies = []
    for k, reg in enumerate(g["regions"]):
        tv, qv = str(tmp_path / ("t%d.vcf" % k)), str(tmp_path / ("q%d.vcf" % k))
        write_text(tv, vcf_text("mock_chr1", [(p, a0, a1, z) for p, a0, a1, _t, z in reg["truth"]]))
        write_text(qv, vcf_text("mock_chr1", [(p, a0, a1, z) for p, a0, a1, _t, z in reg["query"]]))
        feed = feeder.feed_compare(tv, qv, bd, genome)
        b = feed.batch
        assert b.n_regions == 1 and int(b.start[0]) == 0 and int(b.end[0]) == len(g["contig"])
        assert [int(x) for x in b.var_type] == [VT[v[3]] for v in reg["truth"] + reg["query"]]
        assert [int(x) for x in b.var_zyg] == [ZYG[v[4]] for v in reg["truth"] + reg["query"]]
        res = oracle_lib.compare_batch(oracle, b, [c for c in genome.contigs()])
        check_region_expectations(res, 0, reg, b)
        tallies.append(res.tally.copy())
    # and the summary of all of them, written by the library, against the restatement
    total = np.sum(tallies, axis=0).astype(np.uint64)
    out = str(tmp_path / "summary.tsv")
    feeder.write_summary(out, total, "golden", feeder.METRIC_GT | feeder.METRIC_BASEPAIR | feeder.METRIC_HAP)
    text = open(out).read()
    assert text == fo.summary_text(total, "golden", ("GT", "BASEPAIR", "HAP"))
    rows = [l.split("\t") for l in text.splitlines()]
    assert rows[0][:5] == ["compare_label", "comparison", "region_label", "filter", "variant_type"] and len(rows[0]) == 16
    gt_all = rows[1]
    assert gt_all[:5] == ["golden", "GT", "ALL", "ALL", "ALL"]
    want_tp = sum(r["expect"]["gt"][0] for r in g["regions"])
    want_fn = sum(r["expect"]["gt"][1] for r in g["regions"])
    assert int(gt_all[6]) == want_tp and int(gt_all[7]) == want_fn and int(gt_all[5]) == want_tp + want_fn
    assert float(gt_all[11]) == want_tp / (want_tp + want_fn)


def test_summary_float_text_and_csv(tmp_path):
    """ryu's shortest round-trip float text as the csv crate writes it; empty categories are left out; .csv is comma separated"""
    assert [fo.ryu(x) for x in (1.0, 0.5, 1 / 3, 2 / 3, 0.1, 1e-5, 9.5e-6, 123456.0, 1e16, 1.5e-7, float("nan"))] == \
        ["1.0", "0.5", "0.3333333333333333", "0.6666666666666666", "0.1", "0.00001", "9.5e-6", "123456.0", "1e16", "1.5e-7", "NaN"]
    rng = np.random.default_rng(3)
    for trial in range(6):
        tally = np.zeros(288, np.uint64)
        for g in rng.choice(13, size=int(rng.integers(1, 8)), replace=False):
            tally[g * 22:(g + 1) * 22] = rng.integers(0, 10 ** int(rng.integers(1, 9)), size=22)
        if trial == 0:  # a block whose recall and precision are both 0: F1 is 0/0
            tally[:22] = 0
            tally[F["GT_TRUTH_FN"]] = 3
            tally[F["GT_QUERY_FP"]] = 2
        for ext, delim in ((".tsv", "\t"), (".csv", ",")):
            out = str(tmp_path / ("s%d%s" % (trial, ext)))
            feeder.write_summary(out, tally, "lab,el" if ext == ".csv" else "label", 31)
            want = fo.summary_text(tally, '"lab,el"' if ext == ".csv" else "label", ("GT", "BASEPAIR", "HAP", "WEIGHTED_HAP", "RECORD_BP"), delim)
            assert open(out).read() == want
        if trial == 0:
            assert "NaN" in open(out).read()


# ------------------------------------------------------------------ annotated VCFs: BGZF + tabix, read back with plain Python
def bgzf_blocks(path):
    """[(file offset, decompressed bytes)] of every BGZF block; checks the BC extra field and the EOF marker"""
    import struct
    import zlib
    data = open(path, "rb").read()
    out, at = [], 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04" and data[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
        raw = zlib.decompress(data[at + 18:at + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", data, at + bsize - 8)
        assert isize == len(raw) and crc == (zlib.crc32(raw) & 0xFFFFFFFF) and len(raw) <= 0xff00
        out.append((at, raw))
        at += bsize
    assert out[-1][1] == b"" and len(data) - out[-1][0] == 28  # the BGZF end-of-file block
    return out


def reg2bins(beg, end):
    end -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins += list(range(base + (beg >> shift), base + (end >> shift) + 1))
    return bins


def tabix_fetch(vcf_path, chrom, beg, end):
    """records of vcf_path overlapping [beg, end) of chrom, found THROUGH the .tbi (bins, chunks, linear index, virtual offsets)"""
    import struct
    blocks = bgzf_blocks(vcf_path)
    by_off = {o: raw for o, raw in blocks}
    tbi = b"".join(raw for _, raw in bgzf_blocks(vcf_path + ".tbi"))
    assert tbi[:4] == b"TBI\x01"
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", tbi, 4)
    assert (fmt, col_seq, col_beg, col_end, meta, skip) == (2, 1, 2, 0, ord("#"), 0)
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
            bins[b] = [struct.unpack_from("<QQ", tbi, at + 16 * k) for k in range(n_chunk)]
            at += 16 * n_chunk
        n_intv = struct.unpack_from("<i", tbi, at)[0]
        at += 4
        linear = list(struct.unpack_from("<%dQ" % n_intv, tbi, at))
        at += 8 * n_intv
        if name.decode() == chrom:
            found = (bins, linear)
    if found is None:
        return []
    bins, linear = found
    min_off = linear[min(beg >> 14, len(linear) - 1)] if linear else 0
    # flat text addressed by virtual offset
    offs = sorted(by_off)
    recs = []
    for b in reg2bins(beg, end):
        for cb, ce in bins.get(b, []):
            if ce <= min_off:
                continue
            o, within = cb >> 16, cb & 0xFFFF
            text = b""
            k = offs.index(o)
            pos = within
            while (offs[k] << 16 | pos) < ce:  # walk block by block up to the chunk end
                raw = by_off[offs[k]]
                stop = (ce & 0xFFFF) if offs[k] == ce >> 16 else len(raw)
                text += raw[pos:stop]
                if offs[k] == ce >> 16:
                    break
                k, pos = k + 1, 0
            for line in text.decode().splitlines():
                f = line.split("\t")
                p0 = int(f[1]) - 1
                if f[0] == chrom and p0 < end and p0 + len(f[3]) > beg:
                    recs.append(line)
    return sorted(set(recs), key=lambda l: (int(l.split("\t")[1]), l))


def test_annotated_vcfs_bgzf_and_tabix(tmp_path, oracle):
    """truth.vcf.gz / query.vcf.gz as the reference's VariantCategorizer writes them: header additions, one record per variant of
    every solved region, BGZF blocks, and a tabix index through which every record can be fetched"""
    paths = {}
    for name, text in (("ref.fa", EDGE_FASTA), ("hc.bed", EDGE_BED), ("truth.vcf", EDGE_VCF_T), ("query.vcf", EDGE_VCF_Q)):
        paths[name] = str(tmp_path / name)
        write_text(paths[name], text)
    genome = feeder.Genome(paths["ref.fa"])
    feed = feeder.feed_compare(paths["truth.vcf"], paths["query.vcf"], paths["hc.bed"], genome, truth_sample="S1")
    res = oracle_lib.compare_batch(oracle, feed.batch, genome.contigs())
    res.status = res.status.copy()
    res.status[1] = 3  # pretend one region failed: its variants are left out
    regions, _ = fo.generate_regions(fo.load_calls(paths["truth.vcf"], "S1"), fo.load_calls(paths["query.vcf"], ""), fo.read_bed(paths["hc.bed"]),
                                     fo.read_fasta(paths["ref.fa"]))
    for source, name, inp, sample in ((0, "truth.vcf.gz", "truth.vcf", "S1"), (1, "query.vcf.gz", "query.vcf", "")):
        out = str(tmp_path / name)
        feeder.write_annotated_vcf(out, paths[inp], genome, feed.batch, res, source, sample_name=sample, version="v-test", command_line="cmd --x 1")
        text = b"".join(raw for _, raw in bgzf_blocks(out)).decode()
        assert gzip.open(out, "rt").read() == text  # an ordinary gzip reader sees the same
        lines = text.splitlines()
        meta = [l for l in lines if l.startswith("##")]
        src_meta = [l for l in open(paths[inp]).read().splitlines() if l.startswith("##")]
        # the layout the reference's VCF library gives a header: file format, INFO, FILTER, FORMAT, ALT, contig groups, then the other
        # lines grouped by key; the input's lines keep their text, the additions join their groups
        added = ['##FORMAT=<ID=BD,Number=1,Type=String,Description="Benchmark Decision for call (TP/FP/FN)">',
                 '##FORMAT=<ID=EA,Number=1,Type=Integer,Description="Expected Allele count for this genotype">',
                 '##FORMAT=<ID=OA,Number=1,Type=Integer,Description="Observed Allele count for this genotype">',
                 '##FORMAT=<ID=RI,Number=1,Type=Integer,Description="Region ID for the comparison">']
        group = lambda key: [l for l in src_meta if l.startswith("##%s=<" % key)]
        other = [l for l in src_meta if not l.startswith("##fileformat=") and not any(l.startswith("##%s=<" % k) for k in ("INFO", "FILTER", "FORMAT", "ALT", "contig"))]
        assert meta == [l for l in src_meta if l.startswith("##fileformat=")] + group("INFO") + group("FILTER") + group("FORMAT") + added + group("ALT") + \
            group("contig") + other + ['##aardvark_version="v-test"', '##aardvark_command="cmd --x 1"']
        assert lines[len(meta)] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + ("S1" if source == 0 else "Q")
        records = lines[len(meta) + 1:]
        want = fo.annotated_vcf_records(regions, source, res.status, res.var_expected, res.var_observed, res.var_class)
        assert records == want and len(records) > 3
        assert not any(r.split(":")[-1] == str(regions[1]["region_id"]) for r in records)
        # every record comes back through the index, and region queries return exactly the overlapping ones
        for chrom in ("chrA", "chrB", "chrUnused"):
            mine = [r for r in records if r.split("\t")[0] == chrom]
            assert tabix_fetch(out, chrom, 0, 1 << 29) == sorted(set(mine), key=lambda l: (int(l.split("\t")[1]), l))
            for beg, end in ((0, 20), (11, 12), (60, 160), (140, 141), (299, 300), (300, 400)):
                hit = [r for r in mine if int(r.split("\t")[1]) - 1 < end and int(r.split("\t")[1]) - 1 + len(r.split("\t")[3]) > beg]
                assert tabix_fetch(out, chrom, beg, end) == sorted(set(hit), key=lambda l: (int(l.split("\t")[1]), l))


def test_output_header_is_regrouped_like_the_reference_library_writes_it(tmp_path, oracle):
    """a header in an arbitrary order comes out as: file format, INFO, FILTER, FORMAT, ALT, contig definitions, then the other lines
    grouped by key in order of first appearance; a FORMAT definition with the ID of an added one is replaced in place; structured lines are written
    again field by field (fixed field order per kind, "other" fields quoted, IDX last), lines that do not parse are kept as they are"""
    paths = {}
    for name, text in (("ref.fa", EDGE_FASTA), ("hc.bed", EDGE_BED), ("truth.vcf", EDGE_VCF_T), ("query.vcf", EDGE_VCF_Q)):
        paths[name] = str(tmp_path / name)
        write_text(paths[name], text)
    body = [l for l in EDGE_VCF_T.splitlines() if not l.startswith("##")]
    messy = ['##source=callerA', '##contig=<ID=chrA,length=300>', '##FORMAT=<ID=RI,Number=1,Type=String,Description="an older definition">',
             '##fileformat=VCFv4.2', '##FILTER=<ID=LowQual,Description="low">', '##cmdline=first', '##INFO=<ID=DP,Number=1,Type=Integer,Description="depth">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">', '##source=callerB', '##ALT=<ID=DEL,Description="deletion">',
             '##contig=<ID=chrB,length=300>', '##cmdline=second',
             # definitions the library writes differently from how they came in: fields out of order, bare "other" fields, IDX, escapes
             '##INFO=<ID=AF,Type=Float,Description="allele \\"frequency\\", a\\\\b",Number=A,Source=callerA,Version=3>',
             '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read depth",IDX=7,Note="kept, quoted">',
             '##contig=<ID=chrC,assembly=toy,md5=0123abcd,length=300,URL=http://x/y>', '##FILTER=<Description="q<10",ID=q10>',
             '##SAMPLE=<ID=S1,Assay=WGS,Description="the sample">', '##PEDIGREE=<Name_0=child,Name_1=mother>', '##META=<broken']
    src = str(tmp_path / "messy.vcf")
    write_text(src, "\n".join(messy + body) + "\n")
    genome = feeder.Genome(paths["ref.fa"])
    feed = feeder.feed_compare(src, paths["query.vcf"], paths["hc.bed"], genome, truth_sample="S1")
    res = oracle_lib.compare_batch(oracle, feed.batch, genome.contigs())
    out = str(tmp_path / "truth.vcf.gz")
    feeder.write_annotated_vcf(out, src, genome, feed.batch, res, 0, sample_name="S1", version="v", command_line="c")
    meta = [l for l in gzip.open(out, "rt").read().splitlines() if l.startswith("##")]
    assert meta == ['##fileformat=VCFv4.2', '##INFO=<ID=DP,Number=1,Type=Integer,Description="depth">',
                    '##INFO=<ID=AF,Number=A,Type=Float,Description="allele \\"frequency\\", a\\\\b",Source="callerA",Version="3">',
                    '##FILTER=<ID=LowQual,Description="low">', '##FILTER=<ID=q10,Description="q<10">',
                    '##FORMAT=<ID=RI,Number=1,Type=Integer,Description="Region ID for the comparison">', '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
                    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read depth",Note="kept, quoted",IDX=7>',
                    '##FORMAT=<ID=BD,Number=1,Type=String,Description="Benchmark Decision for call (TP/FP/FN)">',
                    '##FORMAT=<ID=EA,Number=1,Type=Integer,Description="Expected Allele count for this genotype">',
                    '##FORMAT=<ID=OA,Number=1,Type=Integer,Description="Observed Allele count for this genotype">',
                    '##ALT=<ID=DEL,Description="deletion">', '##contig=<ID=chrA,length=300>', '##contig=<ID=chrB,length=300>',
                    '##contig=<ID=chrC,length=300,md5=0123abcd,URL=http://x/y,assembly="toy">',
                    '##source=callerA', '##source=callerB', '##cmdline=first', '##cmdline=second',
                    '##SAMPLE=<ID=S1,Assay="WGS",Description="the sample">', '##PEDIGREE=<Name_0=child,Name_1="mother">', '##META=<broken',
                    '##aardvark_version="v"', '##aardvark_command="c"']


def test_bgzf_writer_splits_large_outputs(tmp_path, oracle):
    """more than one 64 KiB block: virtual offsets in the index point into later blocks"""
    p, contig, want_batch = write_case_files(tmp_path, 4000, 2_000_000)
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_compare(p["t"], p["q"], p["bed"], genome, enable_trimming=False)
    res = oracle_lib.compare_batch(oracle, feed.batch, genome.contigs(), threads=8)
    out = str(tmp_path / "truth.vcf.gz")
    feeder.write_annotated_vcf(out, p["t"], genome, feed.batch, res, 0)
    blocks = bgzf_blocks(out)
    assert len(blocks) > 3
    records = [l for l in b"".join(raw for _, raw in blocks).decode().splitlines() if not l.startswith("#")]
    assert len(records) == int(feed.batch.t_cnt.sum())
    for beg, end in ((0, 2_000_000), (500_000, 500_500), (1_234_000, 1_300_000), (1_999_000, 2_000_000)):
        hit = [r for r in records if int(r.split("\t")[1]) - 1 < end and int(r.split("\t")[1]) - 1 + len(r.split("\t")[3]) > beg]
        assert tabix_fetch(out, "chr20", beg, end) == sorted(set(hit), key=lambda l: (int(l.split("\t")[1]), l))


def test_zlib_and_libdeflate_write_and_read_the_same_text(tmp_path):
    """the feeder compresses and inflates BGZF blocks with libdeflate when the system has it (looked up at run time) and with zlib otherwise (or under AVF_ZLIB=1):
    either writer's file holds the same text, and either reader gets the same calls from either file (a fresh process per setting: the choice is made once)"""
    import subprocess, sys
    p, contig, want_batch = write_case_files(tmp_path, 1500, 800_000)
    script = ("import sys, os; sys.path[:0] = [%r, %r]\n"
              "import numpy as np, oracle_lib\nfrom aardvark_amd import feeder\n"
              "g = feeder.Genome(sys.argv[1]); feed = feeder.feed_compare(sys.argv[2], sys.argv[3], sys.argv[4], g, enable_trimming=False)\n"
              "res = oracle_lib.compare_batch(oracle_lib.load(), feed.batch, g.contigs(), threads=4)\n"
              "feeder.write_annotated_vcf(sys.argv[5], sys.argv[2], g, feed.batch, res, 0)\n"
              "print(feed.batch.n_regions, int(feed.batch.t_cnt.sum()), int(feed.batch.var_pos.astype(np.uint64).sum()))\n") % (os.path.dirname(os.path.abspath(__file__)), ROOT)
    seen = {}
    for name, env, truth in (("ld", {}, p["t"]), ("z", {"AVF_ZLIB": "1"}, p["t"]), ("z_reads_ld", {"AVF_ZLIB": "1"}, str(tmp_path / "ld.vcf.gz")),
                             ("ld_reads_z", {}, str(tmp_path / "z.vcf.gz"))):
        out = str(tmp_path / (name + ".vcf.gz"))
        r = subprocess.run([sys.executable, "-c", script, p["fa"], truth, p["q"], p["bed"], out], capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr
        seen[name] = (r.stdout.strip(), [raw for _, raw in bgzf_blocks(out)])
    records = lambda blocks: [l for l in b"".join(blocks).decode().splitlines() if not l.startswith("##")]
    assert records(seen["ld"][1]) == records(seen["z"][1]) and len(records(seen["ld"][1])) > 1000
    assert seen["ld"][0] == seen["z"][0] == seen["z_reads_ld"][0] == seen["ld_reads_z"][0]


def write_case_files(tmp_path, n_truth=3000, length=1_500_000):
    """a small chr20-shaped SNV+indel call set as FASTA(.gz) + BED + two VCFs; returns (paths, contig, batch the generator clusters)"""
    contig = synth.make_contig(length, 15)
    rng = np.random.default_rng(16)
    bed = synth.make_bed(length, 20, 0.9, rng)
    truth = synth.indel_truth(contig, bed, n_truth, 17)
    query = synth.perturb_query(contig, bed, truth, 18, 30)
    want = synth.cluster_regions(length, bed, truth, query, 50)
    p = {k: str(tmp_path / v) for k, v in dict(fa="c.fa.gz", bed="c.bed", t="t.vcf.gz", q="q.vcf.gz", out="out").items()}
    seq = bytes(contig).decode()
    write_text(p["fa"], ">chr20\n" + "\n".join(seq[i:i + 80] for i in range(0, length, 80)) + "\n", "gz")
    write_text(p["bed"], "".join("chr20\t%d\t%d\n" % (a, b) for a, b in bed))
    for path, cs in ((p["t"], truth), (p["q"], query)):
        write_text(path, vcf_text("chr20", [(int(cs.pos[i]), cs.ref[i].decode(), cs.alt[i].decode(), ZNAME[int(cs.zyg[i])]) for i in range(len(cs))]), "members")
    return p, contig, want


def cli_path():
    return os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_compare")


def test_command_line_tool_fails_loudly_without_a_gpu(tmp_path):
    """there is no CPU path behind the tool either: without a HIP device it stops at context creation"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p, _, _ = write_case_files(tmp_path, 50, 60_000)
    r = subprocess.run([cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "-o", p["out"]], capture_output=True, text=True)
    assert r.returncode == 70 and "cannot create the GPU context" in r.stderr
    assert not os.path.exists(os.path.join(p["out"], "summary.tsv"))
    # the options are saved to the debug folder before anything is computed (src/main.rs:64-82), samples resolved to names
    dbg = str(tmp_path / "dbg")
    r = subprocess.run([cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "-o", p["out"], "--output-debug", dbg, "--take", "7",
                        "--enable-haplotype-metrics", "--compare-label", 'a "quoted" label'], capture_output=True, text=True)
    assert r.returncode == 70
    js = json.load(open(os.path.join(dbg, "cli_settings.json")))
    assert list(js)[:4] == ["aardvark_version", "reference_fn", "truth_vcf_filename", "query_vcf_filename"] and len(js) == 23
    assert js["truth_sample"] == "S1" and js["query_sample"] == "S1" and js["compare_label"] == 'a "quoted" label' and js["take_blocks"] == 7
    assert js["enable_haplotype_scoring"] is True and js["enable_record_basepair_scoring"] is False and js["stratifications"] is None
    assert js["min_variant_gap"] == 50 and js["max_edit_distance"] == 5000 and js["skip_blocks"] == 0 and js["regions"] == p["bed"]


@pytest.mark.gpu
def test_command_line_tool_end_to_end(tmp_path, oracle):
    """FASTA + BED + VCFs -> aardvark_amd_compare on the GPU -> summary.tsv, against generator clustering + oracle + restated writer"""
    import subprocess
    p, contig, want_batch = write_case_files(tmp_path)
    r = subprocess.run([cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "-o", p["out"], "--disable-variant-trimming",
                        "--compare-label", "e2e", "--enable-haplotype-metrics", "--enable-weighted-haplotype-metrics", "--enable-record-basepair-metrics",
                        "--batch-regions", "700"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    res = oracle_lib.compare_batch(oracle, want_batch, [contig], threads=8)
    assert (res.status == 0).all()
    want = fo.summary_text(res.tally, "e2e", ("GT", "BASEPAIR", "HAP", "WEIGHTED_HAP", "RECORD_BP"))
    assert open(os.path.join(p["out"], "summary.tsv")).read() == want
    assert "Solved:error blocks: %d : 0" % want_batch.n_regions in r.stderr
    # the annotated VCFs: one record per variant, decisions as the oracle made them
    regions, _ = fo.generate_regions(fo.load_calls(p["t"], "", False), fo.load_calls(p["q"], "", False), fo.read_bed(p["bed"]), fo.read_fasta(p["fa"]))
    for source, name in ((0, "truth.vcf.gz"), (1, "query.vcf.gz")):
        out = os.path.join(p["out"], name)
        records = [l for l in gzip.open(out, "rt").read().splitlines() if not l.startswith("#")]
        assert records == fo.annotated_vcf_records(regions, source, res.status, res.var_expected, res.var_observed, res.var_class)
        assert tabix_fetch(out, "chr20", 700_000, 800_000) == [r for r in records if 700_000 < int(r.split("\t")[1]) + len(r.split("\t")[3]) - 1 and int(r.split("\t")[1]) - 1 < 800_000]


@pytest.mark.gpu
def test_command_line_tool_with_several_contexts(tmp_path):
    """--devices 0,0: two contexts (here on the one GPU of the box) draw the region batches; every output file is byte-identical to the
    single-context run, with and without stratification labels"""
    import subprocess
    p, contig, want_batch = write_case_files(tmp_path)
    beds = {"lo": [(0, 700_000)], "hi": [(600_000, 1_500_000)]}
    for name, iv in beds.items():
        write_text(str(tmp_path / (name + ".bed")), "".join("chr20\t%d\t%d\n" % x for x in iv))
    write_text(str(tmp_path / "strat.tsv"), "".join("%s\t%s.bed\n" % (n, n) for n in beds))
    base = [cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "--disable-variant-trimming"]
    for extra in ([], ["-s", str(tmp_path / "strat.tsv")]):
        outs = []
        for k, dev in enumerate((["--device", "0"], ["--devices", "0,0", "--batch-regions", "400"], ["--devices", "0,0,0"], ["--device", "0", "--batch-form", "wide"],
                                 ["--device", "0", "--batch-regions", "333"], ["--devices", "0,0", "--batch-form", "wide"])):
            out = str(tmp_path / ("out_%d_%d" % (len(extra), k)))
            r = subprocess.run(base + ["-o", out, "-v"] + extra + dev, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert "Solved:error blocks: %d : 0" % want_batch.n_regions in r.stderr
            # several contexts on the packed feed without labels: the job is cut by the library's one rule, hash(region_id) % contexts
            assert ("sharded by hash(region_id)" in r.stderr) == ("--devices" in dev and not extra and "wide" not in dev), r.stderr
            outs.append(out)
        for name in ("summary.tsv", "truth.vcf.gz", "query.vcf.gz"):
            ref = gzip.open(os.path.join(outs[0], name), "rb").read() if name.endswith(".gz") else open(os.path.join(outs[0], name), "rb").read()
            ref = b"\n".join(l for l in ref.split(b"\n") if not l.startswith(b"##aardvark_command"))
            for o in outs[1:]:
                got = gzip.open(os.path.join(o, name), "rb").read() if name.endswith(".gz") else open(os.path.join(o, name), "rb").read()
                assert b"\n".join(l for l in got.split(b"\n") if not l.startswith(b"##aardvark_command")) == ref, (name, o)


@pytest.mark.gpu
def test_command_line_tool_shards_a_selection_of_regions(tmp_path):
    """--skip / --take with --devices 0,0: the shards are cut from the selected regions, results land where the single-context run puts them"""
    import subprocess
    p, contig, want_batch = write_case_files(tmp_path)
    base = [cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "--disable-variant-trimming", "--skip", "57", "--take", "901"]
    outs = []
    for k, dev in enumerate((["--device", "0"], ["--devices", "0,0"], ["--devices", "0,0,0,0"])):
        out = str(tmp_path / ("sel_%d" % k))
        r = subprocess.run(base + ["-o", out] + dev, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(out)
    for name in ("summary.tsv", "truth.vcf.gz", "query.vcf.gz"):
        rd = lambda o: b"\n".join(l for l in (gzip.open(os.path.join(o, name), "rb").read() if name.endswith(".gz") else open(os.path.join(o, name), "rb").read()).split(b"\n")
                                  if not l.startswith(b"##aardvark_command"))
        assert rd(outs[1]) == rd(outs[0]) and rd(outs[2]) == rd(outs[0]), name


# ------------------------------------------------------------------ stratifications: pinned by the reference's own test + fixture
STRAT_TSV = os.path.join(ROOT, "tests", "golden", "example_stratification", "strat.tsv")


def test_example_stratification_known_answers():
    """src/parsing/stratifications.rs:214-243 (test_example_stratification) on the reference's fixture files
    (test_data/example_stratification, copied to tests/golden/): the C++ library and the Python restatement"""
    strat = feeder.Stratifications(STRAT_TSV)
    ostrat = fo.load_stratifications(STRAT_TSV)
    assert len(strat.labels) == 2 and strat.labels == [l for l, _ in ostrat] == ["example1", "example2"]
    assert [strat.n_intervals(0, "mock"), strat.n_intervals(0, "mock2"), strat.n_intervals(1, "mock"), strat.n_intervals(1, "mock2")] == [2, 1, 2, 2]
    for impl in (strat.containments, lambda c, a, b: fo.containments(ostrat, c, a, b)):
        assert impl("mock", 9, 9) == []
        assert impl("mock", 10, 10) == [0]
        assert impl("mock", 14, 14) == [0]
        assert impl("mock", 15, 15) == [0, 1]
        assert impl("mock", 20, 20) == [1]
        assert impl("mock", 25, 25) == [0]
        assert impl("mock", 10, 19) == [0]
        assert impl("mock", 15, 24) == [1]
    for impl in (strat.overlaps, lambda c, a, b: fo.overlaps(ostrat, c, a, b)):
        assert impl("mock", 10, 19) == [0, 1]
        assert impl("mock", 15, 24) == [0, 1]
    # every single-base and short range of the fixture: library == restatement
    for chrom in ("mock", "mock2", "absent"):
        for a in range(0, 40):
            for b in range(a, min(a + 12, 40)):
                assert strat.containments(chrom, a, b) == fo.containments(ostrat, chrom, a, b)
                assert strat.overlaps(chrom, a, b) == fo.overlaps(ostrat, chrom, a, b)
    strat.close()


def test_stratified_summary_and_region_labels(tmp_path, oracle):
    """per-label tallies = sums over the regions a label contains (summary.rs:146-163), written after the ALL block"""
    p, contig, _ = write_case_files(tmp_path, 600, 300_000)
    # three stratification BEDs over chr20: two overlapping halves and scattered small windows; labels sort as a, b, c
    rng = np.random.default_rng(4)
    beds = {"b_right": [(140_000, 300_000)], "a_left": [(0, 160_000)],
            "c_windows": sorted((int(s), int(s) + int(w)) for s, w in zip(rng.integers(0, 299_000, 400), rng.integers(1, 900, 400)))}
    for name, iv in beds.items():
        write_text(str(tmp_path / (name + ".bed")), "".join("chr20\t%d\t%d\n" % x for x in iv))
    write_text(str(tmp_path / "strat.tsv"), "".join("%s\t%s.bed\n" % (n, n) for n in beds))
    strat = feeder.Stratifications(str(tmp_path / "strat.tsv"))
    ostrat = fo.load_stratifications(str(tmp_path / "strat.tsv"))
    assert strat.labels == ["a_left", "b_right", "c_windows"]
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_compare(p["t"], p["q"], p["bed"], genome, enable_trimming=False)
    regions, _ = fo.generate_regions(fo.load_calls(p["t"], "", False), fo.load_calls(p["q"], "", False), fo.read_bed(p["bed"]), fo.read_fasta(p["fa"]))
    res = oracle_lib.compare_batch(oracle, feed.batch, genome.contigs(), threads=4)
    blocks = np.zeros((3, 288), np.uint64)
    seen = set()
    for r, reg in enumerate(regions):
        labels = strat.region_labels(genome, feed.batch, r)
        assert labels == fo.region_labels(ostrat, reg)
        seen.update(labels)
        for l in labels:
            blocks[l, :286] += res.group_metrics[r].reshape(-1).astype(np.uint64)
    assert seen == {0, 1, 2}
    # the whole batch at once (what avk_label_tallies takes), and a window of it
    off, idx = strat.batch_labels(genome, feed.batch)
    assert [idx[int(off[r]):int(off[r + 1])].tolist() for r in range(len(regions))] == [fo.region_labels(ostrat, reg) for reg in regions]
    off2, idx2 = strat.batch_labels(genome, feed.batch, 50, 100)
    assert off2.size == 101 and np.array_equal(idx2, idx[int(off[50]):int(off[150])]) and np.array_equal(off2, off[50:151] - off[50])
    out = str(tmp_path / "summary.tsv")
    feeder.write_summary_stratified(out, res.tally, strat, blocks, "strat", 31)
    want = fo.summary_text(res.tally, "strat", ("GT", "BASEPAIR", "HAP", "WEIGHTED_HAP", "RECORD_BP"), strat_blocks=[(l, blocks[k]) for k, l in enumerate(strat.labels)])
    text = open(out).read()
    assert text == want
    assert [l.split("\t")[2] for l in text.splitlines()[1:] if l.split("\t")[1] == "GT" and l.split("\t")[4] == "ALL"] == ["ALL", "a_left", "b_right", "c_windows"]


@pytest.mark.gpu
def test_command_line_tool_with_stratifications(tmp_path, oracle):
    """--stratification: per-label blocks summed from the per-region metric blocks the GPU returns"""
    import subprocess
    p, contig, want_batch = write_case_files(tmp_path, 2500, 1_200_000)
    beds = {"lowhalf": [(0, 600_000)], "highhalf": [(500_000, 1_200_000)], "tiles": [(k * 10_000, k * 10_000 + 4_000) for k in range(120)]}
    for name, iv in beds.items():
        write_text(str(tmp_path / (name + ".bed")), "".join("chr20\t%d\t%d\n" % x for x in iv))
    write_text(str(tmp_path / "strat.tsv"), "".join("%s\t%s.bed\n" % (n, n) for n in beds))
    r = subprocess.run([cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "-o", p["out"], "--disable-variant-trimming",
                        "-s", str(tmp_path / "strat.tsv"), "--batch-regions", "900", "--output-debug", str(tmp_path / "dbg")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ostrat = fo.load_stratifications(str(tmp_path / "strat.tsv"))
    regions, _ = fo.generate_regions(fo.load_calls(p["t"], "", False), fo.load_calls(p["q"], "", False), fo.read_bed(p["bed"]), fo.read_fasta(p["fa"]))
    res = oracle_lib.compare_batch(oracle, want_batch, [contig], threads=8)
    blocks = np.zeros((3, 288), np.uint64)
    for k, reg in enumerate(regions):
        for l in fo.region_labels(ostrat, reg):
            blocks[l, :286] += res.group_metrics[k].reshape(-1).astype(np.uint64)
    want = fo.summary_text(res.tally, "compare", ("GT", "BASEPAIR"), strat_blocks=[(l, blocks[i]) for i, (l, _) in enumerate(ostrat)])
    assert open(os.path.join(p["out"], "summary.tsv")).read() == want
    assert blocks.sum() > 0
    # --output-debug: the per-region tables, written batch by batch
    full = oracle_lib.compare_batch(oracle, want_batch, [contig], sequences=True, threads=8)
    assert gzip.open(str(tmp_path / "dbg" / "region_summary.tsv.gz"), "rt").read() == fo.region_summary_text(regions, full.status, full.group_metrics)
    seqs = [[full.sequence(k, j).decode() for j in range(5)] for k in range(want_batch.n_regions)]
    assert gzip.open(str(tmp_path / "dbg" / "region_sequences.tsv.gz"), "rt").read() == fo.region_sequences_text(regions, full.status, seqs)


@pytest.mark.gpu
def test_label_tallies_on_the_device(tmp_path, oracle):
    """avk_label_tallies: 20 labels (two launches of 16), per-label sums from the metric blocks that stay on the GPU, against sums of the
    oracle's blocks; then the tool without --output-debug (the path that uses it), batches of 900 regions"""
    import subprocess
    import torch
    torch.cuda.init()
    import aardvark_amd
    p, contig, want_batch = write_case_files(tmp_path, 2500, 1_200_000)
    rng = np.random.default_rng(8)
    names = ["s%02d" % i for i in range(20)]
    for i, name in enumerate(names):
        iv = sorted((int(s), int(s) + int(w)) for s, w in zip(rng.integers(0, 1_190_000, 30 + 10 * i), rng.integers(200, 40_000, 30 + 10 * i)))
        write_text(str(tmp_path / (name + ".bed")), "".join("chr20\t%d\t%d\n" % x for x in iv))
    write_text(str(tmp_path / "strat.tsv"), "".join("%s\t%s.bed\n" % (n, n) for n in names))
    strat = feeder.Stratifications(str(tmp_path / "strat.tsv"))
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_compare(p["t"], p["q"], p["bed"], genome, enable_trimming=False)
    res = oracle_lib.compare_batch(oracle, feed.batch, genome.contigs(), threads=8)
    res.status = res.status.copy()
    off, idx = strat.batch_labels(genome, feed.batch)
    blocks = np.zeros((20, 288), np.uint64)
    for r in range(feed.batch.n_regions):
        for l in idx[int(off[r]):int(off[r + 1])]:
            blocks[int(l), :286] += res.group_metrics[r].reshape(-1).astype(np.uint64)
    assert (blocks.sum(axis=1) > 0).all()
    ctx = aardvark_amd.Context(0)
    ctx.upload_reference(genome.contigs())
    rb = ctx.upload(feed.batch)
    with pytest.raises(aardvark_amd.AardvarkAmdError, match="emit_group_metrics"):
        ctx.set_option("emit_group_metrics", 0)
        ctx.compare_resident(rb)
        ctx.label_tallies(rb, 20, off, idx)
    rb.free()
    ctx.set_option("emit_group_metrics", 1)
    rb = ctx.upload(feed.batch)
    ctx.compare_resident(rb)
    got = ctx.label_tallies(rb, 20, off, idx)
    assert np.array_equal(got, blocks)
    assert np.array_equal(ctx.label_tallies(rb, 20, off, idx, out=got), 2 * blocks)  # sums are added to `out`
    with pytest.raises(aardvark_amd.AardvarkAmdError, match="label index"):
        ctx.label_tallies(rb, 3, off, idx)
    rb.free()
    ctx.close()
    r = subprocess.run([cli_path(), "-r", p["fa"], "-t", p["t"], "-q", p["q"], "-b", p["bed"], "-o", p["out"], "--disable-variant-trimming",
                        "-s", str(tmp_path / "strat.tsv"), "--batch-regions", "900"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = fo.summary_text(res.tally, "compare", ("GT", "BASEPAIR"), strat_blocks=[(l, blocks[i]) for i, l in enumerate(strat.labels)])
    assert open(os.path.join(p["out"], "summary.tsv")).read() == want


def test_debug_tables(tmp_path, oracle):
    """region_summary.tsv.gz / region_sequences.tsv.gz (--output-debug): BGZF text equal to the restated writers"""
    p, contig, _ = write_case_files(tmp_path, 500, 250_000)
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_compare(p["t"], p["q"], p["bed"], genome, enable_trimming=False)
    regions, _ = fo.generate_regions(fo.load_calls(p["t"], "", False), fo.load_calls(p["q"], "", False), fo.read_bed(p["bed"]), fo.read_fasta(p["fa"]))
    res = oracle_lib.compare_batch(oracle, feed.batch, genome.contigs(), sequences=True, threads=4)
    res.status = res.status.copy()
    res.status[3] = 7  # a failed region is left out of both tables
    rs, sq = str(tmp_path / "region_summary.tsv.gz"), str(tmp_path / "region_sequences.tsv.gz")
    feeder.write_debug_tables(rs, sq, genome, feed.batch, res, 31)
    for path in (rs, sq):
        assert b"".join(raw for _, raw in bgzf_blocks(path)).decode() == gzip.open(path, "rt").read()
    assert gzip.open(rs, "rt").read() == fo.region_summary_text(regions, res.status, res.group_metrics, ("GT", "BASEPAIR", "HAP", "WEIGHTED_HAP", "RECORD_BP"))
    seqs = [[res.sequence(r, k).decode() for k in range(5)] for r in range(feed.batch.n_regions)]
    assert gzip.open(sq, "rt").read() == fo.region_sequences_text(regions, res.status, seqs)
    assert len(gzip.open(sq, "rt").read().splitlines()) == feed.batch.n_regions  # header + all but one
