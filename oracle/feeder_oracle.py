"""CPU restatement of the reference's feeder and summary writer in plain Python.  TEST INFRASTRUCTURE: only tests/
may import it; the product is aardvark_amd/csrc/feeder/avf_feeder.cpp.

Parity unpinned: the reference holds no tests or fixtures for src/parsing/ and src/writers/ (0 #[test] functions,
"TODO: we likely need to add tests" at region_generation.rs:814-821), so this restatement is anchored on the source
only; the end-to-end checks that ARE pinned: the 8 solve_compare_region known-answer regions run through
VCF -> feeder -> solver (tests/test_feeder.py), and the worked examples of the reference's documentation (labeled VCF,
summary rows, the two debug tables; tests/test_docs_examples.py), which the C++ writers reproduce byte for byte.

file:line citations are into PacificBiosciences/aardvark v0.10.5.
"""
import gzip
from decimal import Decimal

VT = {n: i for i, n in enumerate(["Snv", "Insertion", "Deletion", "Indel", "SvInsertion", "SvDeletion", "SvDuplication",
                                  "SvInversion", "SvBreakend", "TrContraction", "TrExpansion", "Unknown"])}
ZYG = {n: i for i, n in enumerate(["Unknown", "HomozygousReference", "UnphasedHeterozygous", "PhasedHet01", "PhasedHet10", "HomozygousAlternate"])}


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "rt")


def read_fasta(path):
    """ReferenceGenome::from_fasta (src/main.rs:94): [(name, sequence)] in file order."""
    out = []
    with _open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line.startswith(">"):
                out.append([line[1:].split()[0] if line[1:].split() else "", []])
            else:
                out[-1][1].append(line)
    return [(n, "".join(s)) for n, s in out]


def read_bed(path):
    """LoadedBed::preload_bed_file (noodles_helper.rs:48-86): {chrom: [(start1, end1)]} in order of first appearance,
    intervals 1-based inclusive, sorted by (start, end)."""
    bed = {}
    with _open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line or line.startswith(("#", "track", "browser")):
                continue
            c, s, e = line.split("\t")[:3]
            bed.setdefault(c, []).append((int(s) + 1, int(e)))
    for c in bed:
        bed[c].sort()
    return bed


def parse_genotype(gt):
    """parse_genotype (region_generation.rs:660-712) -> [(alt_index, zygosity name)]"""
    phased = "|" in gt
    alleles = gt.replace("|", "/").split("/")
    if len(alleles) == 1:
        alleles = alleles * 2  # hemizygous is treated as homozygous
    if len(alleles) != 2:
        raise ValueError("allele.len() != [1, 2]: %s" % gt)
    i1, i2 = [0 if a == "." else int(a) for a in alleles]
    if i1 == i2:
        return [(i1, "HomozygousAlternate")] if i1 else []
    out = []
    if i1:
        out.append((i1, "PhasedHet10" if phased else "UnphasedHeterozygous"))
    if i2:
        out.append((i2, "PhasedHet01" if phased else "UnphasedHeterozygous"))
    return out


def variant_type(info, ref, alt):
    """get_variant_type (:715-758) + the constructors' checks (variants.rs:104-383); None = skipped kind"""
    if "SVTYPE" in info and info["SVTYPE"] is not None:
        sv = info["SVTYPE"]
        if sv in ("BND", "DUP"):
            return None
        if sv == "DEL":
            if len(ref) <= 1 or len(alt) > len(ref):
                raise ValueError("bad SV deletion")
            return "SvDeletion"
        if sv == "INS":
            if len(alt) < len(ref) or not ref:
                raise ValueError("bad SV insertion")
            return "SvInsertion"
        raise ValueError("Unsupported SVTYPE detected: %s" % sv)
    if info.get("TRID"):
        return "TrContraction" if len(alt) < len(ref) else "TrExpansion"
    if not ref or not alt:
        raise ValueError("cannot have alleles with 0 length")
    if len(ref) == 1 and len(alt) == 1:
        return "Snv"
    if len(ref) == 1:
        return "Insertion"
    if len(alt) == 1:
        return "Deletion"
    return "Indel"


def load_calls(path, sample, enable_trimming=True):
    """parse_variant over every record (:563-654) -> {chrom: [call dict]} in file order"""
    calls, col, rec = {}, None, 0
    with _open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line.startswith("#"):
                if line.startswith("#CHROM"):
                    hdr = line.split("\t")
                    col = 9 if not sample else hdr.index(sample)
                continue
            fld = line.split("\t")
            this_rec, rec = rec, rec + 1
            chrom, pos1, ref, alts_s, info_s, fmt = fld[0], int(fld[1]), fld[3], fld[4], fld[7], fld[8].split(":")
            if "GT" not in fmt:
                raise ValueError("Missing GT")
            sv = fld[col].split(":")
            gi = fmt.index("GT")
            if gi >= len(sv) or sv[gi] in (".", ""):
                continue
            alts = [] if alts_s in (".", "") else alts_s.split(",")
            info = {}
            if info_s not in (".", ""):
                for kv in info_s.split(";"):
                    k, _, v = kv.partition("=")
                    info[k] = v if "=" in kv else None
            for alt_index, zyg in parse_genotype(sv[gi]):
                alt = alts[alt_index - 1]
                if alt == "*" or alt.startswith("<"):
                    continue
                r, a = ref, alt
                raw = max(len(r), len(a))
                while enable_trimming and len(r) > 1 and len(a) > 1 and r[-1] == a[-1]:
                    r, a = r[:-1], a[:-1]
                if len(r) > 10000 or len(a) > 10000:
                    continue
                vt = variant_type(info, r, a)
                if vt is None:
                    continue
                calls.setdefault(chrom, []).append(dict(pos=pos1 - 1, a0=r, a1=a, raw=raw, type=vt, zyg=zyg, record=this_rec, alt_index=alt_index))
    return calls


def generate_multi_regions(calls_list, bed, contigs, gap=50):
    """RegionIterator::next (:281-478) over k inputs -> list of region dicts (region_id, contig index, chrom, start, end,
    inputs = one call list per input) and the per-input loaded counts."""
    k = len(calls_list)
    names = [n for n, _ in contigs]
    regions, next_id, loaded = [], 0, [0] * k
    for chrom, intervals in bed.items():
        if chrom not in names:
            raise ValueError("Chromosome %s was not found in reference genome" % chrom)
        ci = names.index(chrom)
        chrom_length = len(contigs[ci][1])
        zb_start, zb_end = intervals[0][0] - 1, intervals[-1][1]
        joint = []
        for inp, calls in enumerate(calls_list):
            for c in calls.get(chrom, []):
                last = c["pos"] + len(c["a0"]) - 1
                if zb_start <= c["pos"] < zb_end and zb_start <= last < zb_end:  # is_variant_contained (:764-778)
                    joint.append((inp, c))
                    loaded[inp] += 1
        joint.sort(key=lambda t: t[1]["pos"])  # stable
        head = 0
        for (s1, e1) in intervals:
            ib, ie = s1 - 1, e1
            variants = [[] for _ in range(k)]
            ws = we = None
            while head < len(joint):
                inp, c = joint[head]
                vs, ve = c["pos"], c["pos"] + len(c["a0"])
                if vs < ib:
                    head += 1
                    continue
                if vs >= ie:
                    break
                head += 1
                if ve > ie:
                    continue
                if we is not None and vs >= we:
                    regions.append(dict(region_id=next_id, contig=ci, chrom=chrom, start=ws, end=we, inputs=variants))
                    next_id += 1
                    variants = [[] for _ in range(k)]
                    ws = None
                if ws is None:
                    ws = max(vs - gap, 0)
                flank_end = min(vs + len(c["a0"]) + gap, chrom_length)
                we = flank_end if we is None else max(we, flank_end)
                variants[inp].append(c)
            if ws is not None and we is not None:
                regions.append(dict(region_id=next_id, contig=ci, chrom=chrom, start=ws, end=we, inputs=variants))
                next_id += 1
    return regions, loaded


def generate_regions(truth_calls, query_calls, bed, contigs, gap=50):
    """The compare iterator (:61-122): the same walk over (truth, query) -> region dicts with truth / query lists."""
    multi, loaded = generate_multi_regions([truth_calls, query_calls], bed, contigs, gap)
    return [dict(region_id=m["region_id"], contig=m["contig"], chrom=m["chrom"], start=m["start"], end=m["end"], truth=m["inputs"][0], query=m["inputs"][1])
            for m in multi], loaded


def ryu(x):
    """Text of an f64 as Rust's ryu (the csv crate) writes it."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    if x == 0:
        return "0.0"
    sign, digits, exp = Decimal(repr(x)).as_tuple()
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
        exp += 1
    ds = "".join(map(str, digits))
    k, kk = exp, len(ds) + exp
    if 0 <= k and kk <= 16:
        body = ds + "0" * k + ".0"
    elif 0 < kk <= 16:
        body = ds[:kk] + "." + ds[kk:]
    elif -5 < kk <= 0:
        body = "0." + "0" * (-kk) + ds
    else:
        body = ds[0] + ("." + ds[1:] if len(ds) > 1 else "") + "e" + str(kk - 1)
    return ("-" if sign else "") + body


FIELDS = ["GT_TRUTH_TP", "GT_TRUTH_FN", "GT_QUERY_TP", "GT_QUERY_FP", "GT_TRUTH_FN_GT", "GT_QUERY_FP_GT", "HAP_TRUTH_TP", "HAP_TRUTH_FN", "HAP_QUERY_TP",
          "HAP_QUERY_FP", "WHAP_TRUTH_TP", "WHAP_TRUTH_FN", "WHAP_QUERY_TP", "WHAP_QUERY_FP", "BP_TRUTH_TP", "BP_TRUTH_FN", "BP_QUERY_TP", "BP_QUERY_FP",
          "RBP_TRUTH_TP", "RBP_TRUTH_FN", "RBP_QUERY_TP", "RBP_QUERY_FP"]


def load_stratifications(tsv_path):
    """Stratifications::from_tsv_batch (src/parsing/stratifications.rs:30-86): [(label, {chrom: [(first, last)]})], labels sorted,
    intervals 0-based inclusive"""
    import os
    folder = os.path.dirname(tsv_path)
    files = {}
    for row in open(tsv_path).read().splitlines():
        if not row:
            continue
        label, fn = row.split("\t")[:2]
        if label in files:
            raise ValueError("Duplicate label found: %s" % label)
        files[label] = fn if os.path.isabs(fn) else os.path.join(folder, fn)
    return [(label, {c: [(s - 1, e - 1) for s, e in iv] for c, iv in read_bed(files[label]).items()}) for label in sorted(files)]


def containments(strat, chrom, first, last):
    """:101-113 + :189-199: labels with an interval i.first <= first && i.last >= last"""
    return [k for k, (_l, trees) in enumerate(strat) if any(a <= first and b >= last for a, b in trees.get(chrom, []))]


def overlaps(strat, chrom, first, last):
    return [k for k, (_l, trees) in enumerate(strat) if any(a <= last and b >= first for a, b in trees.get(chrom, []))]


def region_labels(strat, region):
    """solve_compare_region's query (waffle_solver.rs:151-166) with CompareRegion::var_coordinates (compare_region.rs:54-66)"""
    t, q = region["truth"], region["query"]
    start = min([v[0]["pos"] for v in (t, q) if v])
    end = max([v[-1]["pos"] + len(v[-1]["a0"]) for v in (t, q) if v])
    return containments(strat, region["chrom"], start, end - 1)


def summary_text(tally, compare_label="compare", metrics=("GT", "BASEPAIR"), delim="\t", strat_blocks=()):
    """SummaryWriter::write_summary (src/writers/summary.rs:163-395): the ALL block, then one block per stratification label
    (strat_blocks = [(label, tally)])."""
    types = list(VT)
    base = {"GT": 0, "HAP": 6, "WEIGHTED_HAP": 10, "BASEPAIR": 14, "RECORD_BP": 18}
    joints = [("JointIndel", ["Insertion", "Deletion", "Indel"]),
              ("JointStructuralVariant", ["SvInsertion", "SvDeletion", "SvDuplication", "SvInversion", "SvBreakend"]),
              ("JointTandemRepeat", ["TrExpansion", "TrContraction"])]
    lines = [delim.join(["compare_label", "comparison", "region_label", "filter", "variant_type", "truth_total", "truth_tp", "truth_fn", "query_total",
                         "query_tp", "query_fp", "metric_recall", "metric_precision", "metric_f1", "truth_fn_gt", "query_fp_gt"])]

    state = {"tally": tally, "region": "ALL"}

    def group(g):
        return [int(x) for x in state["tally"][g * 22:(g + 1) * 22]]

    def row(kind, vtype, m, extra):
        ttot, qtot = m[0] + m[1], m[2] + m[3]
        rec = m[0] / ttot if ttot else None
        pre = m[2] / qtot if qtot else None
        if rec is not None and pre is not None:
            f1 = 2.0 * rec * pre / (rec + pre) if rec + pre != 0 else float("nan")
        else:
            f1 = None
        cells = [compare_label, kind, state["region"], "ALL", vtype, ttot, m[0], m[1], qtot, m[2], m[3], "" if rec is None else ryu(rec), "" if pre is None else ryu(pre),
                 "" if f1 is None else ryu(f1)] + (list(extra) if kind == "GT" else ["", ""])
        lines.append(delim.join(str(c) for c in cells))

    blocks = [("ALL", tally)] + list(strat_blocks)
    for region, block in blocks:
        state["tally"], state["region"] = block, region
        for kind in metrics:
            b = base[kind]
            g0 = group(0)
            row(kind, "ALL", g0[b:b + 4], g0[4:6])
            for t, name in enumerate(types):
                g = group(1 + t)
                if sum(g[b:b + 4]) == 0:
                    continue
                row(kind, name, g[b:b + 4], g[4:6])
            for label, members in joints:
                s, sx = [0, 0, 0, 0], [0, 0]
                for name in members:
                    g = group(1 + VT[name])
                    s = [a + c for a, c in zip(s, g[b:b + 4])]
                    sx = [a + c for a, c in zip(sx, g[4:6])]
                if sum(s) == 0:
                    continue
                row(kind, label, s, sx)
    return "\n".join(lines) + "\n"


def annotated_vcf_records(regions, source, status, var_expected, var_observed, var_class):
    """The record lines of VariantCategorizer::write_variants (src/writers/variant_categorizer.rs:160-226) for regions as
    generate_regions returns them; the per-variant arrays are indexed like the flattened batch (truth then query per region)."""
    gt = {"Unknown": ".", "HomozygousReference": "0/0", "UnphasedHeterozygous": "0/1", "PhasedHet01": "0|1", "PhasedHet10": "1|0", "HomozygousAlternate": "1/1"}
    cls = ["UNK", "TP", "FN", "FP"]
    lines, v = [], 0
    for r, reg in enumerate(regions):
        for side, key in enumerate(("truth", "query")):
            for c in reg[key]:
                if side == source and status[r] == 0:
                    lines.append("%s\t%d\t.\t%s\t%s\t.\t.\t.\tGT:BD:EA:OA:RI\t%s:%s:%d:%d:%d" % (
                        reg["chrom"], c["pos"] + 1, c["a0"], c["a1"], gt[c["zyg"]], cls[int(var_class[v])], int(var_expected[v]), int(var_observed[v]), reg["region_id"]))
                v += 1
    return lines


def region_summary_text(regions, status, group_metrics, metrics=("GT", "BASEPAIR")):
    """RegionSummaryWriter (src/writers/region_summary.rs:11-139): the joint metrics of every solved region"""
    base = {"GT": 0, "HAP": 6, "WEIGHTED_HAP": 10, "BASEPAIR": 14, "RECORD_BP": 18}
    lines = ["\t".join(["region_id", "coordinates", "comparison", "truth_total", "truth_tp", "truth_fn", "query_total", "query_tp", "query_fp", "metric_recall",
                        "metric_precision", "metric_f1", "truth_fn_gt", "query_fp_gt"])]
    for r, reg in enumerate(regions):
        if status[r] != 0:
            continue
        joint = [int(x) for x in group_metrics[r][0]]
        coords = "%s:%d-%d" % (reg["chrom"], reg["start"] + 1, reg["end"])
        for kind in metrics:
            m = joint[base[kind]:base[kind] + 4]
            ttot, qtot = m[0] + m[1], m[2] + m[3]
            rec = m[0] / ttot if ttot else None
            pre = m[2] / qtot if qtot else None
            f1 = None if rec is None or pre is None else (2.0 * rec * pre / (rec + pre) if rec + pre != 0 else float("nan"))
            cells = [reg["region_id"], coords, kind, ttot, m[0], m[1], qtot, m[2], m[3], "" if rec is None else ryu(rec), "" if pre is None else ryu(pre),
                     "" if f1 is None else ryu(f1)] + ([joint[4], joint[5]] if kind == "GT" else ["", ""])
            lines.append("\t".join(str(c) for c in cells))
    return "\n".join(lines) + "\n"


def region_sequences_text(regions, status, sequences):
    """RegionSequenceWriter (src/writers/region_sequence.rs:9-77); sequences[r] = the 5 strings of region r"""
    lines = ["\t".join(["region_id", "coordinates", "ref_seq", "truth_seq1", "truth_seq2", "query_seq1", "query_seq2"])]
    for r, reg in enumerate(regions):
        if status[r] == 0:
            lines.append("\t".join([str(reg["region_id"]), "%s:%d-%d" % (reg["chrom"], reg["start"] + 1, reg["end"])] + list(sequences[r])))
    return "\n".join(lines) + "\n"
