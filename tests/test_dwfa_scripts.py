"""The reference's DWFA known-answer tests (src/dwfa/dynamic_wfa.rs:283-468, the doc-test of src/dwfa/mod.rs, the 5,278-edit vector of
test_big_early_termination) on the DEVICE aligners: the lane-group aligner of the wave-per-region kernels (engine 0) and the 2-bit
aligner of the lane-per-region kernel (engine 1), through avk_dwfa_script_batch — on CPU through the lane emulator (same source), with
-m gpu on the MI355X through the C-ABI — plus random scripts against the oracle's DWFALite."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import emu_lib
from aardvark_amd import dwfa

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return json.load(open(os.path.join(GOLD, name)))


def case_script(c):
    """a golden case as (baseline, other, steps) + the list of (step index, expected ed) checks"""
    mode = c["mode"]
    if mode == "script":
        b = max((s[1] for s in c["steps"]), key=len).encode()
        o = max((s[2] for s in c["steps"]), key=len).encode()
        steps, checks = [], []
        for k, (op, sb, so, ed) in enumerate(c["steps"]):
            assert b.startswith(sb.encode()) and o.startswith(so.encode())
            steps.append((op, len(sb), len(so)))
            checks.append((k, ed))
        return (b, o, steps), checks, None
    b, o = c["baseline"].encode(), c["other"].encode()
    steps, checks = [], []
    if mode == "finalize":
        steps.append(("finalize", len(b), len(o)))
    elif mode == "update_prefixes":
        for l in range(len(o)):
            steps.append(("update", len(b), l + 1))
            if "ed_each" in c:
                checks.append((len(steps) - 1, c["ed_each"]))
    elif mode == "update_full":
        steps.append(("update", len(b), len(o)))
    if "ed_after_updates" in c and steps:
        checks.append((len(steps) - 1, c["ed_after_updates"]))
    if c.get("then_finalize"):
        steps.append(("finalize", len(b), len(o)))
    if "ed_final" in c and steps:
        checks.append((len(steps) - 1, c["ed_final"]))
    return (b, o, steps), checks, c.get("wavefront")


def acgt(script):
    return all(ch in b"ACGT" for ch in script[0] + script[1])


def golden_scripts():
    out = []
    for c in gold("dwfa.json")["cases"]:
        if c["mode"] == "none":
            continue
        out.append((c["name"],) + case_script(c))
    big = gold("dwfa_big.json")
    b, o = big["baseline"].encode(), big["other"].encode()
    steps = [("update", len(b), i + 1) for i in range(len(o))] + [("finalize", len(b), len(o))]
    checks = [(len(o) - 1, big["ed_after_updates"]), (len(o), big["ed_after_finalize"])]
    out.append(("test_big_early_termination", (b, o, steps), checks, None))
    return out


def check_golden(run, engine):
    cases = [g for g in golden_scripts() if engine == 0 or (acgt(g[1]) and max(len(g[1][0]), len(g[1][1])) <= 192)]
    assert len(cases) >= (13 if engine == 0 else 9)
    res = run([g[1] for g in cases], engine)
    for (name, script, checks, wavefront), (ed, st, wf) in zip(cases, res):
        assert all(s == 0 for s in st), (name, st)
        for k, want in checks:
            assert ed[k] == want, (name, k, ed[k], want)
        if name == "test_big_early_termination":
            assert max(ed[:-1]) <= gold("dwfa_big.json")["max_ed_during_updates"]
        if wavefront is not None:
            assert wf == wavefront, name
        assert len(wf) == 2 * ed[-1] + 1


def random_scripts(seed, n, max_len, alphabet=b"ACGT"):
    rng = np.random.default_rng(seed)
    scripts = []
    for _ in range(n):
        lb = int(rng.integers(0, max_len + 1))
        b = bytes(rng.choice(list(alphabet), size=lb).astype(np.uint8))
        o = bytearray(b)
        for _ in range(int(rng.integers(0, 6))):  # a few edits
            if not o:
                o = bytearray(bytes(rng.choice(list(alphabet), size=1).astype(np.uint8)))
                continue
            p = int(rng.integers(0, len(o)))
            k = int(rng.integers(0, 3))
            if k == 0:
                o[p] = int(rng.choice(list(alphabet)))
            elif k == 1:
                del o[p:p + int(rng.integers(1, 4))]
            elif len(o) < max_len:
                o[p:p] = bytes(rng.choice(list(alphabet), size=min(int(rng.integers(1, 4)), max_len - len(o))).astype(np.uint8))
        o = bytes(o)[:max_len]
        steps, nb, no = [], 0, 0
        while nb < len(b) or no < len(o):  # both strings grow by appends, an update after every growth (haplotype_dwfa.rs:46-67)
            nb = min(len(b), nb + int(rng.integers(0, 40)))
            no = min(len(o), no + int(rng.integers(0, 40)))
            steps.append(("update", nb, no))
        steps.append(("finalize", len(b), len(o)))
        if rng.random() < 0.2:
            steps.append(("update", len(b), len(o)))  # AlreadyFinalized
        scripts.append((b, o, steps))
    return scripts


def oracle_run(oracle, scripts):
    from test_oracle_golden import Dwfa
    out = []
    for b, o, steps in scripts:
        d = Dwfa(oracle)
        ed, st = [], []
        for kind, nb, no in steps:
            rc = d.update(b[:nb], o[:no]) if kind == "update" else d.finalize(b[:nb], o[:no])
            st.append(rc)
            ed.append(d.ed)
        out.append((ed, st, d.wavefront))
    return out


def emu_run(scripts, engine):
    lib = emu_lib.load()
    lib.emu_dwfa_script_batch.argtypes = dwfa.ARGTYPES
    return dwfa.run_scripts(lib.emu_dwfa_script_batch, [], scripts, engine)


@pytest.mark.parametrize("engine", [0, 1])
def test_reference_dwfa_vectors_on_the_kernel_source(engine):
    check_golden(emu_run, engine)


@pytest.mark.parametrize("engine,max_len,alphabet", [(0, 300, b"ACGT"), (0, 90, b"ACGTNacgt"), (1, 190, b"ACGT"), (1, 60, b"AC")])
def test_random_scripts_against_the_oracle_aligner(oracle, engine, max_len, alphabet):
    scripts = random_scripts(7 + engine, 120, max_len, alphabet)
    assert emu_run(scripts, engine) == oracle_run(oracle, scripts)


@pytest.fixture(scope="module")
def ctx():
    import aardvark_amd
    c = aardvark_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine", [0, 1])
def test_reference_dwfa_vectors_on_the_device(ctx, engine):
    check_golden(lambda scripts, e: dwfa.device_scripts(ctx, scripts, e), engine)


@pytest.mark.gpu
@pytest.mark.parametrize("engine,max_len,alphabet,n", [(0, 2000, b"ACGT", 300), (0, 200, b"ACGTNacgt", 2000), (1, 190, b"ACGT", 4000), (1, 60, b"AC", 4000)])
def test_random_scripts_on_the_device(ctx, oracle, engine, max_len, alphabet, n):
    scripts = random_scripts(70 + engine, n, max_len, alphabet)
    assert dwfa.device_scripts(ctx, scripts, engine) == oracle_run(oracle, scripts)
