"""Wall-clock of a full `merge` run on the GPU box: writes K synthetic chr20 call sets of one sample (the base set of BASELINE
configs[1] and K-1 perturbed copies) as FASTA.gz + BED + VCF.gz files, runs aardvark_amd/bin/aardvark_amd_merge on them and checks
every region's classification and the summary against the oracle's pair results + the restated decision and writer."""
import gzip, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from aardvark_amd import feeder, synth
from aardvark_amd._abi import ZYG
from aardvark_amd.merge import MergeConfig, pair_batch
import feeder_oracle as fo
import merge_oracle as mo
import oracle_lib

GT = {ZYG["HomozygousAlternate"]: "1/1", ZYG["UnphasedHeterozygous"]: "0/1", ZYG["PhasedHet01"]: "0|1", ZYG["PhasedHet10"]: "1|0"}
n_truth = int(os.environ.get("N_TRUTH", "50000"))
K = int(os.environ.get("N_INPUTS", "4"))
indel = os.environ.get("INDEL", "0") == "1"
d = tempfile.mkdtemp(prefix="avk_merge_", dir="/tmp")
t0 = time.time()
contig = synth.make_contig(synth.CHR20_LEN, 20250101)
rng = np.random.default_rng(20250101 + 7)
bed = synth.make_bed(synth.CHR20_LEN, 1000, 0.9, rng)
base = (synth.indel_truth if indel else synth.snv_truth)(contig, bed, n_truth, 20250101 + 11)
sets = [base] + [synth.perturb_query(contig, bed, base, 20250102 + i, 500) for i in range(1, K)]
seq = contig.tobytes()
with gzip.open(os.path.join(d, "chr20.fa.gz"), "wb", compresslevel=1) as f:
    f.write(b">chr20\n")
    f.write(b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n")
open(os.path.join(d, "hc.bed"), "w").write("".join("chr20\t%d\t%d\n" % (a, b) for a, b in bed))
hdr = "##fileformat=VCFv4.2\n##contig=<ID=chr20>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n"
vcfs = []
for i, cs in enumerate(sets):
    vcfs.append(os.path.join(d, "in%d.vcf.gz" % i))
    with gzip.open(vcfs[-1], "wt", compresslevel=1) as f:
        f.write(hdr)
        for j in range(len(cs)):
            f.write("chr20\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s\n" % (int(cs.pos[j]) + 1, cs.ref[j].decode(), cs.alt[j].decode(), GT[int(cs.zyg[j])]))
print("fixtures written to %s in %.1f s" % (d, time.time() - t0), flush=True)
summary = os.path.join(d, "summary.tsv")
cmd = [os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_merge"), "-r", os.path.join(d, "chr20.fa.gz")] + [x for v in vcfs for x in ("-i", v)] + \
      ["-b", os.path.join(d, "hc.bed"), "-o", os.path.join(d, "out"), "--output-summary", summary, "--merge-strategy", "all"]
for rep in range(2):
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    print("run %d: exit %d, wall %.2f s" % (rep, r.returncode, time.time() - t0))
    print("\n".join(r.stderr.strip().splitlines()[-4:]))
# the check: pairs from the C oracle, decision and writers from the restatement
genome = feeder.Genome(os.path.join(d, "chr20.fa.gz"))
feed = feeder.feed_merge(vcfs, os.path.join(d, "hc.bed"), genome)
mb = feed.batch
t0 = time.time()
regions = mb.regions()
batch, owner = pair_batch(regions)
st, ex = oracle_lib.optimize_pairs(oracle_lib.load(), batch, genome.contigs(), 50, threads=64)
print("oracle: %d pairs of %d regions in %.2f s" % (batch.n_regions, mb.n_regions, time.time() - t0))
pair = {o: (int(s), int(e)) for o, s, e in zip(owner, st, ex)}
want = []
for m, reg in enumerate(regions):
    if any(pair[(m, i, j)][0] != 0 for i in range(K) for j in range(i + 1, K)):
        want.append(None)
        continue
    want.append(mo.classify([len(v) for v in reg["inputs"]], lambda i, j: pair[(m, i, j)][1], True, True, None))
kinds = {}
for w in want:
    kinds[w[0] if w else "error"] = kinds.get(w[0] if w else "error", 0) + 1
print("classification:", kinds)
calls = [fo.load_calls(v, "") for v in vcfs]
oregions, _ = fo.generate_multi_regions(calls, fo.read_bed(os.path.join(d, "hc.bed")), fo.read_fasta(os.path.join(d, "chr20.fa.gz")))
tags = ["vcf_%d" % i for i in range(K)]
records = [l for l in gzip.open(os.path.join(d, "out", "passing.vcf.gz"), "rt").read().splitlines() if not l.startswith("#")]
print("passing.vcf.gz records identical to oracle + restated writer:", records == mo.passing_vcf_records(oregions, want, tags), len(records))
passing, failed = mo.region_bed_lines(oregions, want)
print("BED files identical:", gzip.open(os.path.join(d, "out", "regions.bed.gz"), "rt").read().splitlines() == passing,
      gzip.open(os.path.join(d, "out", "failed_regions.bed.gz"), "rt").read().splitlines() == failed)
print("summary identical:", open(summary).read() == mo.merge_summary_text(oregions, want, tags, lambda c: mo.TYPE_NAMES.index(c["type"])))
print(open(summary).read())
