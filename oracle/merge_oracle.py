"""CPU restatement of the reference's merge logic around the pairwise exact-match test — TEST INFRASTRUCTURE ONLY
(imported by tests/ only; the product path never touches it).

  classify              solve_merge_region's decision on top of the pair matrix   (src/merge_solver.rs:110-200)
  delta_length          variant_delta_length                                       (src/merge_solver.rs:202-241)
  passing_vcf_records,
  region_bed_lines      VariantMerger::write_results / write_variants / write_region (src/writers/variant_merger.rs:160-311)
  merge_summary_text    MergeSummaryWriter                                          (src/writers/merge_summary.rs)

Pinned by the reference's own known answers for solve_merge_region (tests/golden/merge_solver.json, from
src/merge_solver.rs:243-370).  The writers' text layout has no fixture in the reference's source tree; the record layout of the merged VCF and the BED names are
and the summary table (row order, reason names with their index lists, columns) are pinned by the documentation's examples
(docs/merge.md, tests/test_docs_examples.py).
"""

SIMPLE = {"different": "different", "no_conflict": "no_conflict", "majority": "majority", "conflict_select": "conflict_select", "identical": "identical"}
# derive(Ord) of MergeClassification (src/data_types/merge_benchmark.rs:5-14): declaration order
RANK = {"different": 0, "no_conflict": 1, "majority": 2, "conflict_select": 3, "identical": 4}
TYPE_NAMES = ["Snv", "Insertion", "Deletion", "Indel", "SvInsertion", "SvDeletion", "SvDuplication", "SvInversion", "SvBreakend", "TrContraction",
              "TrExpansion", "Unknown"]
GT_TEXT = {"Unknown": ".", "HomozygousReference": "0/0", "UnphasedHeterozygous": "0/1", "PhasedHet01": "0|1", "PhasedHet10": "1|0", "HomozygousAlternate": "1/1"}


def classify(in_cnt, exact, no_conflict_enabled=False, majority_voting_enabled=False, conflict_selection=None):
    """in_cnt[i] = variants of input i; exact(i, j) for i < j = the pair's exact-match result (after the delta-length rule).
    Returns ("identical",) / ("no_conflict", indices) / ("majority", indices) / ("conflict_select", index) / ("different",)."""
    k = len(in_cnt)
    all_identical, no_conflict = True, True
    match_sets = [{i} for i in range(k)]
    for i in range(k):
        for j in range(i + 1, k):
            e = bool(exact(i, j))
            all_identical &= e
            no_conflict &= in_cnt[i] == 0 or in_cnt[j] == 0 or e
            if e:
                match_sets[i].add(j)
                match_sets[j].add(i)
    maj_count = k // 2 + 1
    first_maj = next((sorted(s) for s in match_sets if len(s) >= maj_count), [])
    if all_identical:
        return ("identical",)
    if no_conflict_enabled and no_conflict:
        return ("no_conflict", [i for i in range(k) if in_cnt[i]])
    if majority_voting_enabled and first_maj:
        return ("majority", first_maj)
    if conflict_selection is not None:
        return ("conflict_select", conflict_selection)
    return ("different",)


def _members(cls, k):
    if cls[0] in ("no_conflict", "majority"):
        return list(cls[1])
    if cls[0] == "conflict_select":
        return [cls[1]]
    if cls[0] == "identical":
        return list(range(k))
    return []


def passing_vcf_records(regions, results, tags):
    """regions: dicts of feeder_oracle.generate_multi_regions; results: one classification tuple (or None = solver error) per region"""
    lines = []
    for reg, cls in zip(regions, results):
        if cls is None or cls[0] == "different":
            continue
        k = len(reg["inputs"])
        src = 0 if cls[0] == "identical" else _members(cls, k)[0]
        sources = ",".join(tags[i] for i in _members(cls, k))
        for c in reg["inputs"][src]:
            lines.append("%s\t%d\t.\t%s\t%s\t.\t.\tSOURCES=%s;MR=%s\tGT:RI\t%s:%d" % (reg["chrom"], c["pos"] + 1, c["a0"], c["a1"], sources, SIMPLE[cls[0]],
                                                                                  GT_TEXT[c["zyg"]], reg["region_id"]))
    return lines


def region_bed_lines(regions, results):
    passing, failed = [], []
    for reg, cls in zip(regions, results):
        if cls is None:
            continue
        line = "%s\t%d\t%d\t%s_%d" % (reg["chrom"], reg["start"], reg["end"], SIMPLE[cls[0]], reg["region_id"])
        (failed if cls[0] == "different" else passing).append(line)
    return passing, failed


def merge_summary_text(regions, results, tags, type_index, delim="\t"):
    """type_index(call) -> position of the call's VariantType in the enum"""
    counts = {}
    for reg, cls in zip(regions, results):
        if cls is None:
            continue
        k = len(reg["inputs"])
        passing = _members(cls, k)
        idx = tuple(cls[1]) if cls[0] in ("no_conflict", "majority") else ((cls[1],) if cls[0] == "conflict_select" else ())
        for i, calls in enumerate(reg["inputs"]):
            for c in calls:
                key = (RANK[cls[0]], idx, type_index(c), i)
                e = counts.setdefault(key, [0, 0, cls[0]])
                e[0 if i in passing else 1] += 1
    if not counts:
        return ""
    out = [delim.join(["merge_reason", "variant_type", "vcf_index", "vcf_label", "pass_variants", "fail_variants"])]
    for key in sorted(counts):
        p, f, name = counts[key]
        reason = SIMPLE[name] + "".join("_%d" % i for i in key[1])
        out.append(delim.join([reason, TYPE_NAMES[key[2]], str(key[3]), tags[key[3]], str(p), str(f)]))
    return "\n".join(out) + "\n"


def classify_k3_majority(pair_status, pair_exact):
    """solve_merge_region's decision (src/merge_solver.rs:149-199) for MANY regions of three inputs at once, majority voting on and the other strategies off, in
    numpy: pair_status / pair_exact [n][3] for the pairs (0,1), (0,2), (1,2) -> (status[n], classification code[n], members bitmask[n]) with the codes of
    include/aardvark_amd.h (0 different, 1 identical, 3 majority).  The whole-genome checks of bench.py and tests/test_gpu_devpack.py use it; `classify` above is
    the general per-region form."""
    import numpy as np
    st = np.asarray(pair_status).reshape(-1, 3)
    ex = np.asarray(pair_exact).reshape(-1, 3) != 0
    err = (st != 0).any(axis=1)
    first_err = np.where(st[:, 0] != 0, st[:, 0], np.where(st[:, 1] != 0, st[:, 1], st[:, 2]))
    ident = ex.all(axis=1) & ~err
    e01, e02, e12 = ex[:, 0].astype(np.int64), ex[:, 1].astype(np.int64), ex[:, 2].astype(np.int64)
    m0, m1, m2 = 1 | (e01 << 1) | (e02 << 2), e01 | 2 | (e12 << 2), e02 | (e12 << 1) | 4
    pc = lambda m: (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1)
    maj = np.where(pc(m0) >= 2, m0, np.where(pc(m1) >= 2, m1, np.where(pc(m2) >= 2, m2, 0)))
    cls = np.where(err, 0, np.where(ident, 1, np.where(maj != 0, 3, 0))).astype(np.uint8)
    members = np.where(err | ident | (maj == 0), 0, maj).astype(np.uint64)
    return np.where(err, first_err, 0).astype(np.int32), cls, members
