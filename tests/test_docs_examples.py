"""The worked examples of the reference's user documentation (tests/golden/docs_examples.json, extracted from docs/compare.md and
docs/merge.md by tests/golden/make_docs_examples.py) as known answers for the writers and the feeder: the reference holds no tests for
src/writers/ and src/parsing/, these printed rows are the only byte-level answers it gives.  The inputs are rebuilt from what the
examples show (positions, alleles, genotypes, the two reference windows of the sequence table), sent through feeder -> solver (the
oracle here, the GPU in the -m gpu variant) -> writers, and the printed rows must come out."""
import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib  # noqa: E402
from aardvark_amd import feeder  # noqa: E402
from aardvark_amd._abi import F  # noqa: E402
from aardvark_amd.merge import MergeResult  # noqa: E402
from test_feeder import write_text  # noqa: E402

DOC = json.load(open(os.path.join(ROOT, "tests", "golden", "docs_examples.json")))


def fields(line):
    f = line.split("\t")
    while f and f[-1] == "":  # the documentation lost trailing tabs of one row
        f.pop()
    return f


def build_compare_inputs(tmp_path):
    """chr1 with the two printed reference windows in place and the printed REF bases at the other example positions"""
    recs = [l.split("\t") for l in DOC["labeled_vcf"][1:]]
    length = 801_400
    seq = bytearray(b"A" * length)
    for row in DOC["region_sequences"][1:]:
        f = row.split("\t")
        start1, end = f[1].split(":")[1].split("-")
        seq[int(start1) - 1:int(end)] = f[2].encode()
    for r in recs:
        pos0 = int(r[1]) - 1
        if seq[pos0:pos0 + 1] != r[3].encode():
            assert not (782955 <= pos0 < 783225)
            seq[pos0:pos0 + 1] = r[3].encode()
    p = {k: str(tmp_path / v) for k, v in dict(fa="doc.fa", bed="doc.bed", t="truth.vcf", q="query.vcf").items()}
    text = bytes(seq).decode()
    write_text(p["fa"], ">chr1 documentation example\n" + "\n".join(text[i:i + 60] for i in range(0, length, 60)) + "\n")
    write_text(p["bed"], "chr1\t0\t%d\n" % length)
    header = "##fileformat=VCFv4.2\n##contig=<ID=chr1>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
    body = "".join("chr1\t%s\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (r[1], r[3], r[4], r[9].split(":")[0]) for r in recs)
    for k in ("t", "q"):
        write_text(p[k], header + DOC["labeled_vcf"][0] + "\n" + body)
    return p


def check_compare_outputs(tmp_path, p, genome, feed, res):
    # labeled VCF: the nine printed records, region ids included
    out = str(tmp_path / "truth.vcf.gz")
    feeder.write_annotated_vcf(out, p["t"], genome, feed.batch, res, 0)
    lines = gzip.open(out, "rt").read().splitlines()
    assert [l for l in lines if not l.startswith("##")] == DOC["labeled_vcf"]
    # the two debug tables: regions 0 and 1 as printed
    feeder.write_debug_tables(str(tmp_path / "region_summary.tsv.gz"), str(tmp_path / "region_sequences.tsv.gz"), genome, feed.batch, res)
    got = gzip.open(str(tmp_path / "region_summary.tsv.gz"), "rt").read().splitlines()
    assert [fields(l) for l in got[:5]] == [fields(l) for l in DOC["region_summary"]]
    got = gzip.open(str(tmp_path / "region_sequences.tsv.gz"), "rt").read().splitlines()
    assert got[:3] == DOC["region_sequences"]


def test_compare_documentation_example(tmp_path):
    p = build_compare_inputs(tmp_path)
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_compare(p["t"], p["q"], p["bed"], genome)
    assert feed.batch.n_regions == 8 and feed.batch.t_cnt.tolist() == [1, 1, 1, 1, 1, 2, 1, 1]
    res = oracle_lib.compare_batch(oracle_lib.load(), feed.batch, genome.contigs(), sequences=True, threads=2)
    check_compare_outputs(tmp_path, p, genome, feed, res)


@pytest.mark.gpu
def test_compare_documentation_example_on_the_gpu(tmp_path):
    import torch
    torch.cuda.init()
    import aardvark_amd
    p = build_compare_inputs(tmp_path)
    genome = feeder.Genome(p["fa"])
    feed = feeder.feed_compare(p["t"], p["q"], p["bed"], genome)
    ctx = aardvark_amd.Context(0)
    ctx.upload_reference(genome.contigs())
    res = ctx.solve_compare_regions(feed.batch, aardvark_amd.CompareConfig(enable_sequences=True))
    check_compare_outputs(tmp_path, p, genome, feed, res)
    ctx.close()


def test_summary_documentation_example(tmp_path):
    """the six printed summary rows: counts in, ratio text out (recall / precision / F1 as the csv crate prints f64)"""
    from aardvark_amd._abi import N_FIELDS, N_GROUPS, TALLY_LEN, VARIANT_TYPES
    rows = [l.split("\t") for l in DOC["summary_tsv"][1:]]
    tally = np.zeros(TALLY_LEN, np.uint64)
    g = tally[:N_GROUPS * N_FIELDS].reshape(N_GROUPS, N_FIELDS)
    group_of = {"ALL": 0, "Snv": 1 + VARIANT_TYPES.index("Snv"), "JointIndel": 1 + VARIANT_TYPES.index("Insertion")}  # the joint row sums the indel types
    for r in rows:
        kind, vt = r[1], r[4]
        pre = "GT_" if kind == "GT" else "BP_"
        grp = g[group_of[vt]]
        grp[F[pre + "TRUTH_TP"]], grp[F[pre + "TRUTH_FN"]], grp[F[pre + "QUERY_TP"]], grp[F[pre + "QUERY_FP"]] = int(r[6]), int(r[7]), int(r[9]), int(r[10])
        if kind == "GT":
            grp[F["GT_TRUTH_FN_GT"]], grp[F["GT_QUERY_FP_GT"]] = int(r[14]), int(r[15])
    out = str(tmp_path / "summary.tsv")
    feeder.write_summary(out, tally, "compare")
    got = {(f[1], f[4]): f for f in (fields(l) for l in open(out).read().splitlines()[1:])}
    assert open(out).read().splitlines()[0] == DOC["summary_tsv"][0]
    for r in rows:
        assert got[(r[1], r[4])] == fields("\t".join(r))


def test_merged_vcf_documentation_example(tmp_path):
    """the nine printed records of passing.vcf.gz: SOURCES lists, merge reasons, GT:RI"""
    from aardvark_amd.merge import MultiBatch
    recs = [l.split("\t") for l in DOC["merged_vcf"][1:]]
    tags = ["pb", "ilmn", "ont"]
    by_region = {}
    for r in recs:
        by_region.setdefault(int(r[9].split(":")[1]), []).append(r)
    regions, st, cls, mem = [], [], [], []
    code = {"different": 0, "identical": 1, "no_conflict": 2, "majority": 3}
    for rid in sorted(by_region):
        rs = by_region[rid]
        info = dict(kv.split("=") for kv in rs[0][7].split(";"))
        members = [tags.index(t) for t in info["SOURCES"].split(",")]
        calls = [(int(r[1]) - 1, r[3], r[4], "Snv", "HomozygousAlternate") for r in rs]
        regions.append({"region_id": rid, "start": int(rs[0][1]) - 51, "end": int(rs[-1][1]) + 50, "inputs": [calls if i in members else [] for i in range(3)]})
        st.append(0)
        cls.append(code[info["MR"]])
        mem.append(sum(1 << i for i in members) if info["MR"] != "identical" else 0)
    mb = MultiBatch.from_regions(regions)
    fa, vcf = str(tmp_path / "m.fa"), str(tmp_path / "first.vcf")
    write_text(fa, ">chr1\n" + "A" * 130_000 + "\n")
    write_text(vcf, "##fileformat=VCFv4.2\n" + DOC["merged_vcf"][0] + "\n")
    genome = feeder.Genome(fa)
    res = MergeResult(np.array(st, np.int32), np.array(cls, np.uint8), np.array(mem, np.uint64), 3)
    out = str(tmp_path / "merged")
    feeder.write_merge_outputs(out, vcf, genome, mb, res, tags=tags)
    lines = gzip.open(os.path.join(out, "passing.vcf.gz"), "rt").read().splitlines()
    assert [l for l in lines if not l.startswith("##")] == DOC["merged_vcf"]
    bed = gzip.open(os.path.join(out, "regions.bed.gz"), "rt").read().splitlines()
    assert [l.split("\t")[3] for l in bed] == ["no_conflict_0", "identical_1", "no_conflict_2", "no_conflict_3", "identical_4", "majority_5", "identical_6", "identical_7"]


def test_merge_summary_documentation_example(tmp_path):
    """the printed merge summary (Snv rows of a three-caller run): one region per merge reason carrying the printed pass / fail counts —
    row order (derive(Ord) of the classification with its index lists), reason names and columns must come out as printed"""
    from aardvark_amd.merge import MultiBatch
    rows = [l.split("\t") for l in DOC["merge_summary"][1:]]
    tags = ["pb", "ilmn", "ont"]
    assert all(r[1] == "Snv" and tags[int(r[2])] == r[3] for r in rows)
    reasons = []
    for r in rows:
        if r[0] not in reasons:
            reasons.append(r[0])
    code = {"different": 0, "identical": 1, "no_conflict": 2, "majority": 3}
    n = len(reasons)
    in_cnt = np.zeros((n, 3), np.uint32)
    cls, mem = np.zeros(n, np.uint8), np.zeros(n, np.uint64)
    for m, reason in enumerate(reasons):
        name = reason.rstrip("_0123456789")
        idx = [int(x) for x in reason[len(name):].split("_") if x]
        cls[m] = code[name]
        mem[m] = sum(1 << i for i in idx)
        for r in rows:
            if r[0] == reason:
                passing, failing = int(r[4]), int(r[5])
                assert (passing == 0) != (failing == 0) and (passing > 0) == (name == "identical" or int(r[2]) in idx)
                in_cnt[m, int(r[2])] = passing + failing
    total = int(in_cnt.sum())
    in_off = np.concatenate([[0], np.cumsum(in_cnt.reshape(-1))[:-1]]).astype(np.uint64)
    zeros = lambda dt: np.zeros(total, dt)
    mb = MultiBatch(3, region_id=np.arange(n), contig_idx=np.zeros(n), start=np.zeros(n), end=np.ones(n), in_off=in_off, in_cnt=in_cnt.reshape(-1),
                    var_pos=zeros(np.uint64), var_type=zeros(np.uint8), var_zyg=np.full(total, 5, np.uint8), var_raw_space=np.ones(total, np.uint32),
                    a0_off=zeros(np.uint64), a0_len=np.ones(total, np.uint32), a1_off=np.ones(total, np.uint64), a1_len=np.ones(total, np.uint32),
                    allele_bytes=np.frombuffer(b"AC", np.uint8))
    res = MergeResult(np.zeros(n, np.int32), cls, mem, 3)
    out = str(tmp_path / "merge_summary.tsv")
    feeder.write_merge_summary(out, mb, res, tags=tags)
    assert open(out).read().splitlines() == DOC["merge_summary"]
