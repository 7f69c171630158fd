"""Seeded region scenarios shared by the emulator (CPU) and GPU parity tests: the reference's
known-answer regions, the synthetic benchmark workloads, and adversarial fuzz regions that reach
the parts no reference test pins (branch quota, auto-fail, incompatible variants, long alleles,
non-ACGT symbols, invalid inputs)."""
import json
import os

import numpy as np

from aardvark_amd import RegionBatch, synth
from aardvark_amd._abi import VT, ZYG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ZY = ["UnphasedHeterozygous", "PhasedHet01", "PhasedHet10", "HomozygousAlternate"]


def golden():
    g = json.load(open(os.path.join(GOLD, "waffle_solver.json")))
    regions = [{"start": r["start"], "end": r["end"], "truth": r["truth"], "query": r["query"]} for r in g["regions"]]
    return [g["contig"].encode()], RegionBatch.from_regions(regions)


def vtype_of(a0, a1):
    if len(a0) == 1 and len(a1) == 1:
        return "Snv"
    if len(a0) == 1:
        return "Insertion"
    if len(a1) == 1:
        return "Deletion"
    return "Indel"


def random_variant(rng, contig, lo, hi, max_len=8, alphabet=b"ACGT"):
    """a random SNV / insertion / deletion / indel whose REF span lies in [lo, hi)"""
    kind = rng.integers(0, 4)
    rl = 1 if kind in (0, 1) else int(rng.integers(2, max_len + 1))
    rl = min(rl, hi - lo)
    pos = int(rng.integers(lo, hi - rl + 1))
    ref = bytes(contig[pos:pos + rl])
    if kind == 0 or (kind == 2 and rl == 1):
        alt = bytes([alphabet[int(rng.integers(0, len(alphabet)))]])
    elif kind == 1:
        alt = ref[:1] + bytes(rng.choice(list(alphabet), size=int(rng.integers(1, max_len + 1))).astype(np.uint8))
    elif kind == 2:
        alt = ref[:1]
    else:
        alt = bytes(rng.choice(list(alphabet), size=int(rng.integers(2, max_len + 1))).astype(np.uint8))
    return (pos, ref, alt, vtype_of(ref, alt), ZY[int(rng.integers(0, 4))])


def fuzz_regions(seed, n, max_vars=5, max_len=8, contig_len=4000, span=(12, 160), related=0.6, alphabet=b"ACGT", repeat_unit=None):
    """random regions; with probability `related` the query is a perturbed copy of the truth"""
    rng = np.random.default_rng(seed)
    if repeat_unit is None:
        contig = np.frombuffer(bytes(rng.choice(list(alphabet), size=contig_len).astype(np.uint8)), dtype=np.uint8).copy()
    else:
        contig = np.frombuffer((repeat_unit * (contig_len // len(repeat_unit) + 1))[:contig_len], dtype=np.uint8).copy()
        # sprinkle a few substitutions so the contig is not perfectly periodic
        idx = rng.integers(0, contig_len, size=contig_len // 40)
        contig[idx] = rng.choice(list(alphabet), size=idx.size).astype(np.uint8)
    regions = []
    for _ in range(n):
        L = int(rng.integers(span[0], span[1]))
        start = int(rng.integers(0, contig_len - L))
        end = start + L
        nt = int(rng.integers(0, max_vars + 1))
        truth = sorted([random_variant(rng, contig, start, end, max_len, alphabet) for _ in range(nt)], key=lambda v: v[0])
        if rng.random() < related and truth:
            query = []
            for v in truth:
                u = rng.random()
                if u < 0.15:
                    continue
                z = v[4] if u < 0.8 else ZY[int(rng.integers(0, 4))]
                if z in ("PhasedHet01", "PhasedHet10") and rng.random() < 0.5:
                    z = "UnphasedHeterozygous"
                query.append((v[0], v[1], v[2], v[3], z))
            for _ in range(int(rng.integers(0, 3))):
                query.append(random_variant(rng, contig, start, end, max_len, alphabet))
            query.sort(key=lambda v: v[0])
        else:
            nq = int(rng.integers(0, max_vars + 1))
            query = sorted([random_variant(rng, contig, start, end, max_len, alphabet) for _ in range(nq)], key=lambda v: v[0])
        regions.append({"start": start, "end": end, "truth": truth, "query": query})
    return [bytes(contig)], RegionBatch.from_regions(regions)


def quota_regions(seed, n=6, n_query=14):
    """many unphased query hets against a sparse truth: 2^n_query orientations >> max_branch_factor,
    so the per-depth quota of optimize_sequences (query_optimizer.rs:222-225) decides the answer"""
    rng = np.random.default_rng(seed)
    contig = synth.ACGT[rng.integers(0, 4, size=3000, dtype=np.uint8)]
    regions = []
    for _ in range(n):
        start = int(rng.integers(0, 2000))
        end = start + 220
        pos = np.sort(rng.choice(np.arange(start + 5, end - 5), size=n_query, replace=False))
        query = []
        for p in pos:
            ref = bytes(contig[p:p + 1])
            alt = bytes(synth._snv_alt(contig[p:p + 1], rng))
            query.append((int(p), ref, alt, "Snv", "UnphasedHeterozygous"))
        truth = [q[:4] + (ZY[int(rng.integers(0, 4))],) for q in query if rng.random() < 0.4]
        regions.append({"start": start, "end": end, "truth": truth, "query": query})
    return [bytes(contig)], RegionBatch.from_regions(regions)


def autofail_regions():
    """single-base deletions spread over a homopolymer, more of them in truth than in query: every
    subset of the same size spells the same haplotype, so the exact-match search
    (exact_gt_optimizer.rs) runs past 500 expansions without a sync point and its auto-fail pruning
    (:309-339) decides which alleles are flipped"""
    contig = b"CG" + b"A" * 120 + b"TC" + b"G" * 40
    regions = []
    for nt, nq, zy in ((8, 5, "HomozygousAlternate"), (10, 6, "HomozygousAlternate"), (10, 4, "UnphasedHeterozygous"), (5, 9, "PhasedHet01")):
        truth = [(4 + 3 * i, b"AA", b"A", "Deletion", zy if zy != "UnphasedHeterozygous" else "PhasedHet10") for i in range(nt)]
        query = [(50 + 3 * i, b"AA", b"A", "Deletion", zy) for i in range(nq)]
        regions.append({"start": 0, "end": len(contig) - 30, "truth": truth, "query": query})
    return [contig], RegionBatch.from_regions(regions)


def long_allele_regions(seed=5):
    """a 3 kbp insertion / deletion pair, a 600 bp SV-typed pair, and a region whose two call sets
    differ by hundreds of edits (wavefronts far beyond the LDS tier)"""
    rng = np.random.default_rng(seed)
    contig = synth.ACGT[rng.integers(0, 4, size=12000, dtype=np.uint8)]
    ins = bytes(synth.ACGT[rng.integers(0, 4, size=3000, dtype=np.uint8)])
    ins2 = bytearray(ins)
    for i in range(0, 3000, 97):
        ins2[i] = ord("A") if ins2[i] != ord("A") else ord("C")
    regions = [
        {"start": 100, "end": 400, "truth": [(200, contig[200:201].tobytes(), contig[200:201].tobytes() + ins, "Insertion", "HomozygousAlternate")],
         "query": [(200, contig[200:201].tobytes(), contig[200:201].tobytes() + bytes(ins2), "Insertion", "UnphasedHeterozygous")]},
        {"start": 1000, "end": 4600, "truth": [(1100, contig[1100:4101].tobytes(), contig[1100:1101].tobytes(), "Deletion", "PhasedHet10")],
         "query": [(1100, contig[1100:4101].tobytes(), contig[1100:1101].tobytes(), "Deletion", "UnphasedHeterozygous"),
                   (4200, contig[4200:4201].tobytes(), b"T" if contig[4200] != ord("T") else b"G", "Snv", "HomozygousAlternate")]},
        {"start": 5000, "end": 5900, "truth": [(5100, contig[5100:5700].tobytes(), contig[5100:5101].tobytes(), "SvDeletion", "HomozygousAlternate")],
         "query": [(5100, contig[5100:5101].tobytes(), contig[5100:5101].tobytes() + ins[:600], "SvInsertion", "HomozygousAlternate")]},
        {"start": 7000, "end": 7800, "truth": [(7050, contig[7050:7051].tobytes(), contig[7050:7051].tobytes() + ins[:500], "TrExpansion", "PhasedHet01")],
         "query": [(7400, contig[7400:7700].tobytes(), contig[7400:7401].tobytes(), "TrContraction", "UnphasedHeterozygous")]},
    ]
    return [contig.tobytes()], RegionBatch.from_regions(regions)


def non_acgt_regions(seed=9):
    """reference and alleles with N, IUPAC and lower-case bytes: compared as raw bytes"""
    return fuzz_regions(seed, 40, max_vars=4, contig_len=2500, alphabet=b"ACGTNacgtRY")


def invalid_regions():
    contig = b"ACGTACGTACGTACGTACGTACGTACGTACGT"
    ok = (10, b"G", b"T", "Snv", "HomozygousAlternate")
    regions = [
        {"start": 0, "end": 20, "truth": [ok], "query": [ok]},                                   # fine
        {"start": 5, "end": 40, "truth": [ok], "query": []},                                      # window past the contig end
        {"start": 0, "end": 20, "truth": [(25, b"C", b"T", "Snv", "HomozygousAlternate")], "query": []},   # variant outside the window
        {"start": 0, "end": 20, "truth": [(12, b"A", b"T", "Snv", "HomozygousAlternate"), ok], "query": []},  # unsorted
        {"start": 0, "end": 20, "truth": [(10, b"G", b"", "Deletion", "HomozygousAlternate")], "query": []},  # empty allele
        {"start": 0, "end": 20, "truth": [(10, b"G", b"T", "Snv", "HomozygousReference")], "query": []},     # assert_eq!(zyg, HomAlt) panics
        {"start": 0, "end": 20, "truth": [(10, b"G", b"T", "Snv", "Unknown")], "query": [ok]},
        {"start": 0, "end": 20, "truth": [(10, b"GTA", b"T", "Deletion", "HomozygousAlternate", 2)], "query": []},  # raw space < allele
        {"start": 0, "end": 20, "truth": [], "query": [], "contig": 3},                            # unknown contig
        {"start": 8, "end": 8, "truth": [], "query": []},                                          # empty window, no variants
        {"start": 0, "end": 32, "truth": [], "query": [ok]},                                       # fine again
    ]
    return [contig], RegionBatch.from_regions(regions)


def chr20_small(n_truth=3000):
    contig, batch = synth.config_chr20_snv(n_truth=n_truth, contig_len=3_000_000, n_intervals=60, n_extra=max(1, n_truth // 100))
    return [contig], batch


def indel_small(n_truth=2500):
    contig, batch = synth.config_indel_mix(n_truth=n_truth, contig_len=1_500_000, n_intervals=40)
    return [contig], batch


def optimizer_golden_regions():
    """the regions of the reference's optimize_sequences tests (src/query_optimizer.rs:533-665) as compare regions, with their expectations"""
    q = json.load(open(os.path.join(GOLD, "query_optimizer.json")))
    regions = [{"start": r["start"], "end": r["end"], "truth": [tuple(v) for v in r["truth"]], "query": [tuple(v) for v in r["query"]]} for r in q["regions"]]
    return [q["contig"].encode()], RegionBatch.from_regions(regions), [r["expect"] for r in q["regions"]]


def max_allele_regions(seed=15):
    """alleles at the reference's size limit (variants over 10 kbp are dropped by its feeder, region_generation.rs:621-626): a 10,000-base
    insertion pair that differs in a hundred places, a 10,000-base deletion against a 9,000-base deletion, and the reference's 5,278-edit
    DWFA vector (dynamic_wfa.rs:453-468) as a region: the window is its 651-base string, the query call replaces it by its 5,929-base string"""
    rng = np.random.default_rng(seed)
    big = json.load(open(os.path.join(GOLD, "dwfa_big.json")))
    base, other = big["baseline"].encode(), big["other"].encode()
    contig = np.concatenate([synth.ACGT[rng.integers(0, 4, size=40000, dtype=np.uint8)], np.frombuffer(base, np.uint8), synth.ACGT[rng.integers(0, 4, size=500, dtype=np.uint8)]])
    ins = bytes(synth.ACGT[rng.integers(0, 4, size=9999, dtype=np.uint8)])
    ins2 = bytearray(ins)
    for i in range(0, 9999, 101):
        ins2[i] = ord("A") if ins2[i] != ord("A") else ord("C")
    a = lambda lo, hi: contig[lo:hi].tobytes()
    regions = [
        {"start": 100, "end": 400, "truth": [(200, a(200, 201), a(200, 201) + ins, "Insertion", "HomozygousAlternate")],
         "query": [(200, a(200, 201), a(200, 201) + bytes(ins2), "Insertion", "UnphasedHeterozygous")]},
        {"start": 1000, "end": 11200, "truth": [(1100, a(1100, 11100), a(1100, 1101), "Deletion", "PhasedHet10")],
         "query": [(1100, a(1100, 10100), a(1100, 1101), "Deletion", "UnphasedHeterozygous"),
                   (11150, a(11150, 11151), b"T" if contig[11150] != ord("T") else b"G", "Snv", "HomozygousAlternate")]},
        {"start": 40000, "end": 40000 + len(base), "truth": [], "query": [(40000, base, other, "Indel", "HomozygousAlternate")]},
    ]
    return [contig.tobytes()], RegionBatch.from_regions(regions), big["ed_after_finalize"]
