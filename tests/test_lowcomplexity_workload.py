"""The low-complexity workload of bench.py (secondary.lowcomplexity_genome): what synth.add_low_complexity writes into a contig, and that the kernels' logic (emulator)
solves a small instance of the workload — calls in and beside simple-repeat tracts, windows next to assembly gaps of N — exactly as the oracle does."""
import numpy as np

import emu_lib
import oracle_lib
from aardvark_amd import synth


def longest_runs(contig):
    """lengths of the maximal single-base runs of a contig"""
    change = np.flatnonzero(np.diff(contig) != 0)
    edges = np.concatenate(([-1], change, [contig.size - 1]))
    return np.diff(edges)


def test_background_has_tracts_gaps_and_a_mask():
    plain = synth.make_contig_fast(30_000_000, 7)
    contig = plain.copy()
    mask, gaps = synth.add_low_complexity(contig, 11)
    assert contig.size == plain.size and mask.dtype == bool and mask.size == contig.size
    assert len(gaps) == 1 and all(10_000 <= b - a <= 50_000 for a, b in gaps)
    for a, b in gaps:
        assert (contig[a:b] == ord("N")).all()
    assert set(np.unique(contig).tolist()) <= set(b"ACGTN")
    changed = float((contig != plain).mean())
    assert 0.015 < changed < 0.06  # tracts cover about 4 % of the bases (a rewritten base equals the old one a quarter of the time)
    assert 0.3 < float(mask.mean()) < 0.7
    runs_plain, runs = longest_runs(plain), longest_runs(contig[contig != ord("N")])
    assert runs_plain.max() < 16 and (runs >= 20).sum() > 1000  # homopolymer tracts of tens of bases that a uniform contig does not have
    again = plain.copy()
    mask2, gaps2 = synth.add_low_complexity(again, 11)
    assert np.array_equal(again, contig) and np.array_equal(mask2, mask) and gaps2 == gaps  # seeded


def test_small_instance_through_the_emulator_equals_oracle(oracle):
    contig, batch = synth.config_indel_mix_v2(n_truth=2500, contig_len=1_000_000, low_complexity=True, str_frac=0.5)
    assert batch.n_regions > 1500
    want = oracle_lib.compare_batch(oracle, batch, [contig], threads=4)
    got = emu_lib.compare_batch(batch, [contig], n_waves=16)
    assert got.diff(want) == []
    assert got.lane_solved > 0.9 * batch.n_regions  # the lane classes still take nearly all of it
