"""Seeded synthetic truth/query call sets and the region clustering that feeds the solver.

The reference ships no VCF/FASTA fixtures, so the benchmark workloads are
generated here: BASELINE.json configs[1] ("Synthetic chr20: 50k SNV-only truth vs query,
confident BED") and a whole-genome-shaped SNV+indel mix.  Regions are formed exactly like the
reference's RegionIterator (src/parsing/region_generation.rs:373-470): variants of both call
sets are merged by position (stable, truth first), a variant joins the open window while
`pos < window_end`, a window is `[first_pos - gap, max(pos + ref_len + gap))` clipped to the
contig, and windows never span BED intervals.
"""
import numpy as np

from ._abi import VT, ZYG, RegionBatch

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
CHR20_LEN = 64_444_167


def make_contig(length, seed):
    rng = np.random.default_rng(seed)
    return ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def make_bed(length, n_intervals, coverage, rng):
    """n non-overlapping sorted intervals covering ~coverage of the contig."""
    edges = np.linspace(0, length, n_intervals + 1).astype(np.int64)
    lo, hi = edges[:-1], edges[1:]
    span = ((hi - lo) * coverage).astype(np.int64)
    slack = (hi - lo) - span
    off = (rng.random(n_intervals) * slack).astype(np.int64)
    return np.stack([lo + off, lo + off + span], axis=1)


class CallSet:
    """A sorted list of bi-allelic calls on one contig: pos, REF/ALT bytes (ragged), type, zygosity."""

    def __init__(self, pos, ref, alt, vtype, zyg, raw_space=None):
        order = np.argsort(pos, kind="stable")
        self.pos = np.asarray(pos, np.int64)[order]
        self.ref = [ref[i] for i in order]
        self.alt = [alt[i] for i in order]
        self.vtype = np.asarray(vtype, np.uint8)[order]
        self.zyg = np.asarray(zyg, np.uint8)[order]
        rl = np.array([len(r) for r in self.ref], np.int64)
        al = np.array([len(a) for a in self.alt], np.int64)
        self.ref_len, self.alt_len = rl, al
        self.raw_space = np.maximum(rl, al) if raw_space is None else np.asarray(raw_space, np.int64)[order]

    def __len__(self):
        return len(self.pos)


def _snv_alt(ref_base, rng):
    """uniform among the three non-REF bases"""
    idx = np.searchsorted(ACGT, ref_base)
    return ACGT[(idx + rng.integers(1, 4, size=ref_base.size)) % 4]


def _random_zyg(n, rng):
    """60 % het (half phased 0|1 / 1|0, half 0/1), 40 % 1/1"""
    u = rng.random(n)
    z = np.full(n, ZYG["HomozygousAlternate"], np.uint8)
    het = u < 0.6
    phased = het & (rng.random(n) < 0.5)
    z[het] = ZYG["UnphasedHeterozygous"]
    z[phased] = np.where(rng.random(int(phased.sum())) < 0.5, ZYG["PhasedHet01"], ZYG["PhasedHet10"])
    return z


def positions_in_bed(bed, n, rng):
    """n distinct positions uniform over the BED-covered bases"""
    lens = bed[:, 1] - bed[:, 0]
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    picks = rng.choice(total, size=n, replace=False) if n * 4 < total else rng.permutation(total)[:n]
    picks.sort()
    k = np.searchsorted(cum, picks, side="right") - 1
    return bed[k, 0] + (picks - cum[k])


def snv_truth(contig, bed, n, seed):
    rng = np.random.default_rng(seed)
    pos = positions_in_bed(bed, n, rng)
    ref = contig[pos]
    alt = _snv_alt(ref, rng)
    return CallSet(pos, [bytes([b]) for b in ref], [bytes([b]) for b in alt], np.full(n, VT["Snv"], np.uint8), _random_zyg(n, rng))


def perturb_query(contig, bed, truth, seed, n_extra, drop=0.01, flip=0.005, change=0.005):
    """query = truth with per-variant drop / zygosity flip / ALT change, all GTs unphased,
    plus n_extra random SNVs (false positives)."""
    rng = np.random.default_rng(seed)
    n = len(truth)
    keep = rng.random(n) >= drop
    pos = truth.pos[keep]
    ref = [truth.ref[i] for i in np.nonzero(keep)[0]]
    alt = [truth.alt[i] for i in np.nonzero(keep)[0]]
    vt = truth.vtype[keep].copy()
    zyg = truth.zyg[keep].copy()
    het = zyg != ZYG["HomozygousAlternate"]
    zyg[het] = ZYG["UnphasedHeterozygous"]
    m = len(pos)
    fl = rng.random(m) < flip
    zyg[fl] = np.where(zyg[fl] == ZYG["HomozygousAlternate"], ZYG["UnphasedHeterozygous"], ZYG["HomozygousAlternate"])
    ch = np.nonzero(rng.random(m) < change)[0]
    for i in ch:
        if vt[i] == VT["Snv"]:
            a = _snv_alt(np.frombuffer(alt[i], np.uint8), rng)  # a base different from the truth ALT
            if bytes(a) == ref[i]:
                a = _snv_alt(np.frombuffer(ref[i] , np.uint8), rng)
                if bytes(a) == alt[i]:
                    continue
            alt[i] = bytes(a)
        else:  # change the last base of the longer allele's tail
            s = bytearray(alt[i])
            s[-1] = int(_snv_alt(np.array([s[-1]], np.uint8), rng)[0])
            if bytes(s) != ref[i]:
                alt[i] = bytes(s)
    if n_extra:
        epos = positions_in_bed(bed, n_extra, rng)
        eref = contig[epos]
        ealt = _snv_alt(eref, rng)
        pos = np.concatenate([pos, epos])
        ref += [bytes([b]) for b in eref]
        alt += [bytes([b]) for b in ealt]
        vt = np.concatenate([vt, np.full(n_extra, VT["Snv"], np.uint8)])
        ez = _random_zyg(n_extra, rng)
        ez[ez != ZYG["HomozygousAlternate"]] = ZYG["UnphasedHeterozygous"]
        zyg = np.concatenate([zyg, ez])
    return CallSet(pos, ref, alt, vt, zyg)


def indel_truth(contig, bed, n, seed, snv_frac=0.82, close_frac=0.03, str_frac=0.05):
    """SNV + insertion + deletion mix (lengths geometric, mean 3, cap 50); a fraction of sites is
    placed within 30 bp of another site; a fraction of indels sits in an injected homopolymer / STR
    context is left to the contig (random sequence has few), so representation shifts come from
    `shift_representation`."""
    rng = np.random.default_rng(seed)
    base = positions_in_bed(bed, n, rng)
    nclose = int(n * close_frac)
    if nclose:
        src = rng.choice(n, size=nclose, replace=False)
        base[src] = np.clip(base[(src + 1) % n] + rng.integers(1, 30, size=nclose), 1, contig.size - 64)
    base = np.unique(base)
    n = base.size
    u = rng.random(n)
    kinds = np.where(u < snv_frac, 0, np.where(u < snv_frac + (1 - snv_frac) / 2, 1, 2))
    lens = np.minimum(rng.geometric(1.0 / 3.0, size=n), 50)
    ref, alt, vt = [], [], np.zeros(n, np.uint8)
    for i in range(n):
        p = int(base[i])
        if kinds[i] == 0:
            r = contig[p:p + 1]
            ref.append(bytes(r))
            alt.append(bytes(_snv_alt(r, rng)))
            vt[i] = VT["Snv"]
        elif kinds[i] == 1:
            ins = ACGT[rng.integers(0, 4, size=int(lens[i]))]
            ref.append(bytes(contig[p:p + 1]))
            alt.append(bytes(contig[p:p + 1]) + bytes(ins))
            vt[i] = VT["Insertion"]
        else:
            L = int(lens[i])
            ref.append(bytes(contig[p:p + 1 + L]))
            alt.append(bytes(contig[p:p + 1]))
            vt[i] = VT["Deletion"]
    return CallSet(base, ref, alt, vt, _random_zyg(n, rng))


def add_multiallelic(calls, frac, seed):
    """a share of the sites is multi-allelic (`1/2`): the site's call becomes a heterozygous one and a second
    heterozygous call with another ALT is added at the same position (what the feeder's multi-ALT split makes of a `1/2` record;
    phased sites get the complementary phase)."""
    rng = np.random.default_rng(seed)
    n = len(calls)
    pick = np.nonzero(rng.random(n) < frac)[0]
    pos, ref, alt, vt, zyg = list(calls.pos), list(calls.ref), list(calls.alt), list(calls.vtype), list(calls.zyg)
    for i in pick:
        r, a = ref[i], alt[i]
        if len(r) == 1 and len(a) == 1:  # SNV: another base
            b = bytes(_snv_alt(np.frombuffer(a, np.uint8), rng))
            if b == r:
                continue
            a2, t2 = b, VT["Snv"]
        elif len(r) == 1:  # insertion: the inserted bases once more
            a2, t2 = a + a[1:], VT["Insertion"]
        else:  # deletion: one base less deleted, or an SNV at the anchor when nothing is left
            a2, t2 = (r[:2], VT["Deletion"]) if len(r) > 2 else (bytes(_snv_alt(np.frombuffer(r[:1], np.uint8), rng)) + r[1:], VT["Indel"])
        phased = rng.random() < 0.5
        zyg[i] = ZYG["PhasedHet01"] if phased else ZYG["UnphasedHeterozygous"]
        pos.append(pos[i]); ref.append(r); alt.append(a2); vt.append(t2)
        zyg.append(ZYG["PhasedHet10"] if phased else ZYG["UnphasedHeterozygous"])
    return CallSet(np.array(pos), ref, alt, np.array(vt, np.uint8), np.array(zyg, np.uint8))


def cluster_regions(contig_len, bed, truth, query, gap=50, contig_idx=0, region_id_base=0):
    """RegionIterator::next for one contig (region_generation.rs:373-470) -> RegionBatch."""
    nt, nq = len(truth), len(query)
    pos = np.concatenate([truth.pos, query.pos])
    rlen = np.concatenate([truth.ref_len, query.ref_len])
    side = np.concatenate([np.zeros(nt, np.int8), np.ones(nq, np.int8)])
    local = np.concatenate([np.arange(nt), np.arange(nq)])
    order = np.argsort(pos, kind="stable")  # joint_vec.sort_by_key(position): truth before query on ties
    pos, rlen, side, local = pos[order], rlen[order], side[order], local[order]
    # containment (get_variant_containment :794-812): start inside an interval and end inside it
    k = np.searchsorted(bed[:, 0], pos, side="right") - 1
    ok = (k >= 0)
    kk = np.clip(k, 0, None)
    ok &= (pos < bed[kk, 1]) & (pos + rlen <= bed[kk, 1])
    pos, rlen, side, local, k = pos[ok], rlen[ok], side[ok], local[ok], k[ok]
    n = pos.size
    if n == 0:
        z = np.zeros(0, np.int64)
        return RegionBatch(z, z, z, z, z, z, z, z, z, z, z, z, z, z, z, z, np.zeros(1, np.uint8))
    flank_end = np.minimum(pos + rlen + gap, contig_len)
    big = np.int64(1) << 40
    seg_max = np.maximum.accumulate(flank_end + k * big) - k * big  # running max restarted per interval
    brk = np.ones(n, bool)
    brk[1:] = (k[1:] != k[:-1]) | (pos[1:] >= seg_max[:-1])  # :397 pos >= window_end
    win = np.cumsum(brk) - 1
    nwin = int(win[-1]) + 1
    first = np.nonzero(brk)[0]
    last = np.concatenate([first[1:], [n]]) - 1
    w_start = np.maximum(pos[first] - gap, 0)  # saturating_sub
    # window end = max flank end inside the window
    w_end = np.maximum.reduceat(flank_end, first)
    # variants of a window: truth block then query block, each in position order
    vorder = np.lexsort((np.arange(n), side, win))
    side_s, local_s, win_s = side[vorder], local[vorder], win[vorder]
    t_cnt = np.bincount(win_s[side_s == 0], minlength=nwin)
    q_cnt = np.bincount(win_s[side_s == 1], minlength=nwin)
    tot = t_cnt + q_cnt
    woff = np.concatenate([[0], np.cumsum(tot)[:-1]])
    t_off, q_off = woff, woff + t_cnt
    is_t = side_s == 0
    vpos = np.where(is_t, truth.pos[np.where(is_t, local_s, 0)], query.pos[np.where(is_t, 0, local_s)])
    vtype = np.where(is_t, truth.vtype[np.where(is_t, local_s, 0)], query.vtype[np.where(is_t, 0, local_s)])
    vzyg = np.where(is_t, truth.zyg[np.where(is_t, local_s, 0)], query.zyg[np.where(is_t, 0, local_s)])
    vraw = np.where(is_t, truth.raw_space[np.where(is_t, local_s, 0)], query.raw_space[np.where(is_t, 0, local_s)])
    refs = [truth.ref[j] if t else query.ref[j] for t, j in zip(is_t, local_s)]
    alts = [truth.alt[j] if t else query.alt[j] for t, j in zip(is_t, local_s)]
    a0_len = np.array([len(x) for x in refs], np.int64)
    a1_len = np.array([len(x) for x in alts], np.int64)
    both = a0_len + a1_len
    base = np.concatenate([[0], np.cumsum(both)[:-1]])
    blob = b"".join(r + a for r, a in zip(refs, alts))
    return RegionBatch(np.arange(nwin) + region_id_base, np.full(nwin, contig_idx), w_start, w_end, t_off, t_cnt, q_off, q_cnt,
                       vpos, vtype, vzyg, vraw, base, a0_len, base + a0_len, a1_len, np.frombuffer(blob, np.uint8))


def config_chr20_snv(n_truth=50_000, contig_len=CHR20_LEN, n_intervals=1000, n_extra=500,
                     seed_ref=20250101, seed_query=20250102, gap=50):
    """BASELINE.json configs[1]: synthetic chr20, 50k SNV-only truth vs query, confident BED."""
    contig = make_contig(contig_len, seed_ref)
    rng = np.random.default_rng(seed_ref + 7)
    bed = make_bed(contig_len, n_intervals, 0.9, rng)
    truth = snv_truth(contig, bed, n_truth, seed_ref + 11)
    query = perturb_query(contig, bed, truth, seed_query, n_extra)
    batch = cluster_regions(contig_len, bed, truth, query, gap)
    return contig, batch


def config_indel_mix(n_truth=200_000, contig_len=CHR20_LEN, n_intervals=1000, seed_ref=20250103, seed_query=20250104, gap=50):
    """A SNV+indel mix on one contig (the per-contig shape of BASELINE.json configs[2])."""
    contig = make_contig(contig_len, seed_ref)
    rng = np.random.default_rng(seed_ref + 7)
    bed = make_bed(contig_len, n_intervals, 0.9, rng)
    truth = indel_truth(contig, bed, n_truth, seed_ref + 11)
    query = perturb_query(contig, bed, truth, seed_query, max(1, n_truth // 100))
    batch = cluster_regions(contig_len, bed, truth, query, gap)
    return contig, batch
