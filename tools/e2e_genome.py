"""Whole-genome-shaped end-to-end run (the stand-in for BASELINE configs[2]): 24 contigs with GRCh38
primary lengths (scaled by SCALE), SNV+indel truth/query call sets at HG002 density, written as FASTA + BED + VCF.gz, then
aardvark_amd_compare on the GPU.  Prints the tool's stage breakdown.  SCALE=1 needs about 12 GB of host memory and 8 GB of disk."""
import gzip, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aardvark_amd import synth
from aardvark_amd._abi import ZYG

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
          114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
GT = {ZYG["HomozygousAlternate"]: "1/1", ZYG["UnphasedHeterozygous"]: "0/1", ZYG["PhasedHet01"]: "0|1", ZYG["PhasedHet10"]: "1|0"}
scale = float(os.environ.get("SCALE", "0.1"))
BGZF = os.environ.get("PLAIN_GZIP", "0") != "1"  # inputs as bgzip writes them (what the reference needs for its tabix queries); PLAIN_GZIP=1: one gzip stream


class BgzfText:
    """text writer producing BGZF: independent gzip members of at most 0xff00 input bytes with the BC extra field, then the end-of-file block"""

    def __init__(self, path):
        import zlib
        self.f, self.buf, self.z = open(path, "wb"), bytearray(), zlib

    def write(self, text):
        self.buf += text.encode()
        while len(self.buf) >= 0xff00:
            self._block(bytes(self.buf[:0xff00]))
            del self.buf[:0xff00]

    def _block(self, chunk):
        import struct
        co = self.z.compressobj(1, self.z.DEFLATED, -15)
        payload = co.compress(chunk) + co.flush()
        self.f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(payload) + 25) + payload +
                     struct.pack("<II", self.z.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

    def close(self):
        if self.buf:
            self._block(bytes(self.buf))
        self._block(b"")
        self.f.close()


def open_vcf(path):
    return BgzfText(path) if BGZF else gzip.open(path, "wt", compresslevel=1)
density = 3.9e6 / sum(GRCH38)  # truth variants per base
d = tempfile.mkdtemp(prefix="avk_genome_", dir=os.environ.get("TMPDIR", "/tmp"))
t0 = time.time()
hdr = "##fileformat=VCFv4.2\n" + "".join("##contig=<ID=%s>\n" % n for n in NAMES) + \
      "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tHG002\n"
FASTA_BGZF = os.environ.get("FASTA_BGZF", "0") == "1"  # the reference as bgzip writes it (genome.fa.gz) instead of plain text


class BgzfBytes:
    def __init__(self, path):
        self.t = BgzfText(path)

    def write(self, b):
        self.t.buf += b
        while len(self.t.buf) >= 0xff00:
            self.t._block(bytes(self.t.buf[:0xff00]))
            del self.t.buf[:0xff00]

    def close(self):
        self.t.close()


FASTA_NAME = "genome.fa.gz" if FASTA_BGZF else "genome.fa"
fa = BgzfBytes(os.path.join(d, FASTA_NAME)) if FASTA_BGZF else open(os.path.join(d, FASTA_NAME), "wb")
bedf = open(os.path.join(d, "hc.bed"), "w")
vt = open_vcf(os.path.join(d, "truth.vcf.gz"))
vq = open_vcf(os.path.join(d, "query.vcf.gz"))
vt.write(hdr); vq.write(hdr)
n_merge = int(os.environ.get("MERGE_INPUTS", "0"))  # also run `merge` with truth + query + (n_merge - 2) more callers of the same sample
vm = [open_vcf(os.path.join(d, "caller%d.vcf.gz" % i)) for i in range(2, n_merge)]
for f in vm:
    f.write(hdr)
n_regions = n_truth = 0
for ci, (name, full) in enumerate(zip(NAMES, GRCH38)):
    length = max(int(full * scale), 200_000)
    # the call sets of bench.py's workload (synth.contig_calls: multi-allelic sites as two records at one position, repeat-run indels with
    # the query record shifted by whole units)
    contig, bed, truth, query = synth.contig_calls(ci, length, density)
    n_truth += len(truth)
    fa.write(b">" + name.encode() + b"\n")
    pad = (-length) % 80
    rows = np.concatenate([contig, np.full(pad, ord("N"), np.uint8)]).reshape(-1, 80)
    body = np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
    fa.write(body[:len(body) - pad - 1] + b"\n" if pad else body)
    bedf.write("".join("%s\t%d\t%d\n" % (name, a, b) for a, b in bed))
    def records(cs):
        out = []
        for i in range(len(cs)):
            a0, a1 = cs.alleles(contig, i)
            out.append("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (name, int(cs.pos[i]) + 1, a0.decode(), a1.decode(), GT[int(cs.zyg[i])]))
        return "".join(out)
    for f, cs in ((vt, truth), (vq, query)):
        f.write(records(cs))
    for i, f in enumerate(vm):
        f.write(records(synth.genome_query(contig, bed, truth, (np.zeros(len(truth), np.int64), np.zeros(len(truth), np.int64)), 20250105 + 1000 * i + ci, max(1, len(truth) // 100))))
    del contig, truth, query, rows, body
for f in [fa, bedf, vt, vq] + vm:
    f.close()
n_strat = int(os.environ.get("STRAT", "0"))  # labelled BED sets for --stratification: label i covers about (i + 1) / (n + 1) of every contig, in 200 intervals per contig
if n_strat:
    with open(os.path.join(d, "strat.tsv"), "w") as ts:
        for i in range(n_strat):
            ts.write("label_%02d\tstrat_%02d.bed\n" % (i, i))
            rng = np.random.default_rng(777 + i)
            with open(os.path.join(d, "strat_%02d.bed" % i), "w") as bf:
                for name, full in zip(NAMES, GRCH38):
                    length = max(int(full * scale), 200_000)
                    for a, b in synth.make_bed(length, 200, (i + 1) / (n_strat + 1), rng):
                        bf.write("%s\t%d\t%d\n" % (name, a, b))
print("fixtures: %d truth variants over %d contigs (scale %.2f) written to %s in %.0f s" % (n_truth, len(NAMES), scale, d, time.time() - t0), flush=True)
cmd = [os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_compare"), "-r", os.path.join(d, FASTA_NAME), "-t", os.path.join(d, "truth.vcf.gz"),
       "-q", os.path.join(d, "query.vcf.gz"), "-b", os.path.join(d, "hc.bed"), "-o", os.path.join(d, "out"), "--disable-variant-trimming"]
if n_strat:
    cmd += ["-s", os.path.join(d, "strat.tsv")]
if os.environ.get("DEVICES"):  # several solver contexts, e.g. DEVICES=0,0 or 0,1,2,3
    cmd += ["--devices", os.environ["DEVICES"]]
if os.environ.get("CLI_ARGS"):  # further options of the tool, e.g. CLI_ARGS="--batch-form wide"
    cmd += os.environ["CLI_ARGS"].split()
for _ in range(int(os.environ.get("RUNS", "1")) - 1):  # RUNS=n: the same command n times, the last one is the one checked below
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    print("exit %d, wall %.2f s; %s" % (r.returncode, time.time() - t0, " | ".join(l for l in r.stderr.strip().splitlines() if l.startswith("stages") or l.startswith("Comparisons"))), flush=True)
t0 = time.time()
r = subprocess.run(cmd, capture_output=True, text=True)
print("exit %d, wall %.2f s" % (r.returncode, time.time() - t0))
print("\n".join(l for l in r.stderr.strip().splitlines() if not l.startswith("Error while solving")))
summary_text = open(os.path.join(d, "out", "summary.tsv")).read()
print("\n".join(l for l in summary_text.splitlines() if "\tALL\tALL\tALL\t" in l or l.startswith("compare_label")))
if os.environ.get("KEEP_SUMMARY"):
    open(os.environ["KEEP_SUMMARY"], "w").write(summary_text)
if os.environ.get("VERIFY", "0") == "1":  # the whole run again on the CPU: feeder -> oracle (all host threads) -> restated summary writer
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import feeder_oracle as fo
    import oracle_lib
    from aardvark_amd import feeder
    t0 = time.time()
    genome = feeder.Genome(os.path.join(d, FASTA_NAME))
    feed = feeder.feed_compare(os.path.join(d, "truth.vcf.gz"), os.path.join(d, "query.vcf.gz"), os.path.join(d, "hc.bed"), genome, enable_trimming=False)
    res = oracle_lib.compare_batch(oracle_lib.load(), feed.batch, genome.contigs(), threads=os.cpu_count())
    want = fo.summary_text(res.tally, "compare", ("GT", "BASEPAIR"))
    print("oracle on %d regions in %.1f s (feeder + %d threads); summary.tsv identical to oracle + restated writer: %s; error regions %d" % (
        feed.batch.n_regions, time.time() - t0, os.cpu_count(), summary_text == want if not n_strat else "n/a (stratified)", int((res.status != 0).sum())))
    # per-variant decisions: every record of the two annotated VCFs (position and the GT:BD:EA:OA:RI column) against the oracle's
    # per-variant arrays (VariantCategorizer::write_variants, src/writers/variant_categorizer.rs:160-226: one record per variant of every
    # solved region, regions in order, the side's variants in order)
    t0 = time.time()
    b = feed.batch
    gt_txt = np.array([".", "0/0", "0/1", "0|1", "1|0", "1/1"])
    cls_txt = np.array(["UNK", "TP", "FN", "FP"])
    ok_all = True
    for side, name, off, cnt in ((0, "truth.vcf.gz", b.t_off, b.t_cnt), (1, "query.vcf.gz", b.q_off, b.q_cnt)):
        cnt64 = np.where(res.status == 0, cnt, 0).astype(np.int64)
        first = np.cumsum(cnt64) - cnt64
        reg = np.repeat(np.arange(b.n_regions), cnt64)
        v = np.repeat(off.astype(np.int64), cnt64) + (np.arange(int(cnt64.sum())) - np.repeat(first, cnt64))
        want_pos = b.var_pos[v].astype(np.int64) + 1
        col = np.char.add(np.char.add(np.char.add(np.char.add(gt_txt[b.var_zyg[v]], ":"), cls_txt[res.var_class[v]]), ":"),
                          np.char.add(np.char.add(res.var_expected[v].astype(str), ":"), np.char.add(np.char.add(res.var_observed[v].astype(str), ":"), b.region_id[reg].astype(str))))
        got_pos, got_col = [], []
        for line in gzip.open(os.path.join(d, "out", name), "rt"):
            if line[0] == "#":
                continue
            f = line.rstrip("\n").split("\t")
            got_pos.append(int(f[1])); got_col.append(f[9])
        same = len(got_pos) == len(want_pos) and np.array_equal(np.array(got_pos), want_pos) and np.array_equal(np.array(got_col), col)
        ok_all = ok_all and same
        print("%s: %d records, positions and GT:BD:EA:OA:RI identical to the oracle's per-variant decisions: %s" % (name, len(got_pos), same))
    print("per-variant verification %s in %.0f s" % ("PASSED" if ok_all else "FAILED", time.time() - t0))
def _cleanup():
    if os.environ.get("KEEP", "0") != "1":  # the fixtures are gigabytes: gone unless KEEP=1
        import shutil
        shutil.rmtree(d, ignore_errors=True)


import atexit
atexit.register(_cleanup)
if n_merge >= 2:  # the shape of BASELINE configs[4] on one GPU: majority vote over the callers
    vcfs = [os.path.join(d, "truth.vcf.gz"), os.path.join(d, "query.vcf.gz")] + [os.path.join(d, "caller%d.vcf.gz" % i) for i in range(2, n_merge)]
    cmd = [os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_merge"), "-r", os.path.join(d, FASTA_NAME)] + [x for v in vcfs for x in ("-i", v)] + \
          ["-b", os.path.join(d, "hc.bed"), "-o", os.path.join(d, "merged"), "--output-summary", os.path.join(d, "merge_summary.tsv"), "--merge-strategy", "majority",
           "--disable-variant-trimming"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    print("merge of %d inputs: exit %d, wall %.2f s" % (n_merge, r.returncode, time.time() - t0))
    print("\n".join(l for l in r.stderr.strip().splitlines() if not l.startswith("Error while solving")))
    print(open(os.path.join(d, "merge_summary.tsv")).read())
    if os.environ.get("VERIFY", "0") == "1" and n_merge == 3:
        # every region's merge reason again on the CPU: k-input feed, the three input pairs through the oracle, the majority rule of
        # solve_merge_region (merge_solver.rs:149-199) with numpy; compared with the names in the two BED files of the tool
        sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_lib
        from aardvark_amd import feeder
        from aardvark_amd._abi import RegionBatch
        t0 = time.time()
        genome = feeder.Genome(os.path.join(d, FASTA_NAME))
        mb = feeder.feed_merge(vcfs, os.path.join(d, "hc.bed"), genome, enable_trimming=False).batch
        n = mb.n_regions
        off, cnt = mb.in_off.reshape(n, 3), mb.in_cnt.reshape(n, 3)
        ex = {}
        for i, j in ((0, 1), (0, 2), (1, 2)):
            pb = RegionBatch(mb.region_id, mb.contig_idx, mb.start, mb.end, off[:, i], cnt[:, i], off[:, j], cnt[:, j], mb.var_pos, mb.var_type, mb.var_zyg,
                             mb.var_raw_space, mb.a0_off, mb.a0_len, mb.a1_off, mb.a1_len, mb.allele_bytes)
            st, e = oracle_lib.optimize_pairs(oracle_lib.load(), pb, genome.contigs(), 50, threads=os.cpu_count())
            assert (st == 0).all()
            ex[(i, j)] = e.astype(bool)
        identical = ex[(0, 1)] & ex[(0, 2)] & ex[(1, 2)]
        # first input whose match set (itself + the inputs it matches) reaches 2 of 3
        maj = np.where(ex[(0, 1)], 1, np.where(ex[(0, 2)], 2, np.where(ex[(1, 2)], 3, 0)))  # 1: {0,1}  2: {0,2}  3: {1,2}  0: none
        want = np.where(identical, "identical", np.where(maj > 0, "majority", "different"))
        got = {}
        for name in ("regions.bed.gz", "failed_regions.bed.gz"):
            for line in gzip.open(os.path.join(d, "merged", name), "rt"):
                reason, rid = line.rstrip("\n").split("\t")[3].rsplit("_", 1)
                got[int(rid)] = reason
        same = len(got) == n and all(got[int(r)] == w for r, w in zip(mb.region_id, want))
        print("merge reasons of %d regions identical to oracle pairs + majority rule: %s (%.1f s; identical %d, majority %d, different %d)" % (
            n, same, time.time() - t0, int(identical.sum()), int(((maj > 0) & ~identical).sum()), int((want == "different").sum())))
if os.environ.get("KEEP", "0") != "1":
    subprocess.run(["rm", "-rf", d])
