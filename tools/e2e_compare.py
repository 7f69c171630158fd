"""Wall-clock of a full `compare` run on the GPU box: writes the synthetic chr20 call sets of BASELINE configs[1]
as FASTA.gz + BED + truth/query VCF.gz files, runs aardvark_amd/bin/aardvark_amd_compare on them and checks summary.tsv against
the oracle's tally of the generator's own regions."""
import gzip, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from aardvark_amd import synth
from aardvark_amd._abi import ZYG
import feeder_oracle as fo
import oracle_lib

GT = {ZYG["HomozygousAlternate"]: "1/1", ZYG["UnphasedHeterozygous"]: "0/1", ZYG["PhasedHet01"]: "0|1", ZYG["PhasedHet10"]: "1|0"}
n_truth = int(os.environ.get("N_TRUTH", "50000"))
d = tempfile.mkdtemp(prefix="avk_e2e_", dir="/tmp")
t0 = time.time()
contig = synth.make_contig(synth.CHR20_LEN, 20250101)
rng = np.random.default_rng(20250101 + 7)
bed = synth.make_bed(synth.CHR20_LEN, 1000, 0.9, rng)
truth = synth.snv_truth(contig, bed, n_truth, 20250101 + 11)
if os.environ.get("MULTIALLELIC", "1") == "1":  # 2 % of the sites carry two ALT alleles, written as one `1/2` record (the feeder splits it again)
    truth = synth.add_multiallelic(truth, 0.02, 20250101 + 13)
query = synth.perturb_query(contig, bed, truth, 20250102, 500)
batch = synth.cluster_regions(synth.CHR20_LEN, bed, truth, query, 50)
seq = contig.tobytes()
fasta_text = b">chr20\n" + b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n"
if os.environ.get("PLAIN_GZIP", "0") == "1":  # one gzip stream: inflated by one thread
    with gzip.open(os.path.join(d, "chr20.fa.gz"), "wb", compresslevel=1) as f:
        f.write(fasta_text)
else:  # as bgzip writes it (what samtools faidx wants): inflated block-parallel
    from test_feeder import bgzf_bytes
    open(os.path.join(d, "chr20.fa.gz"), "wb").write(bgzf_bytes(fasta_text, 0xff00, 1))
open(os.path.join(d, "hc.bed"), "w").write("".join("chr20\t%d\t%d\n" % (a, b) for a, b in bed))
hdr = "##fileformat=VCFv4.2\n##contig=<ID=chr20>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
for name, cs in (("truth.vcf.gz", truth), ("query.vcf.gz", query)):
    with gzip.open(os.path.join(d, name), "wt", compresslevel=1) as f:
        f.write(hdr)
        i, merged = 0, 0
        while i < len(cs):
            two = i + 1 < len(cs) and cs.pos[i + 1] == cs.pos[i] and cs.ref[i + 1] == cs.ref[i] and cs.alt[i + 1] != cs.alt[i]
            z = (int(cs.zyg[i]), int(cs.zyg[i + 1])) if two else None
            if two and z in ((ZYG["UnphasedHeterozygous"],) * 2, (ZYG["PhasedHet01"], ZYG["PhasedHet10"]), (ZYG["PhasedHet10"], ZYG["PhasedHet01"])):
                gt = "1/2" if z[0] == ZYG["UnphasedHeterozygous"] else ("2|1" if z[0] == ZYG["PhasedHet01"] else "1|2")
                f.write("chr20\t%d\t.\t%s\t%s,%s\t.\tPASS\t.\tGT\t%s\n" % (int(cs.pos[i]) + 1, cs.ref[i].decode(), cs.alt[i].decode(), cs.alt[i + 1].decode(), gt))
                i += 2
                merged += 1
                continue
            f.write("chr20\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (int(cs.pos[i]) + 1, cs.ref[i].decode(), cs.alt[i].decode(), GT[int(cs.zyg[i])]))
            i += 1
        print("%s: %d records, %d of them multi-allelic" % (name, len(cs) - merged, merged))
print("fixtures written to %s in %.1f s (%d regions expected)" % (d, time.time() - t0, batch.n_regions), flush=True)
cmd = [os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_compare"), "-r", os.path.join(d, "chr20.fa.gz"), "-t", os.path.join(d, "truth.vcf.gz"),
       "-q", os.path.join(d, "query.vcf.gz"), "-b", os.path.join(d, "hc.bed"), "-o", os.path.join(d, "out")]
for rep in range(2):
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    print("run %d: exit %d, wall %.2f s" % (rep, r.returncode, time.time() - t0))
    print(r.stderr.strip())
res = oracle_lib.compare_batch(oracle_lib.load(), batch, [contig], threads=64)
want = fo.summary_text(res.tally, "compare", ("GT", "BASEPAIR"))
got = open(os.path.join(d, "out", "summary.tsv")).read()
print("summary.tsv identical to oracle + restated writer:", got == want)
print(got)
