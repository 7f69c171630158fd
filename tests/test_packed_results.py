"""The packed result form (avk_result_batch::region_packed / var_packed: 8 bytes per region, 1 per call; include/aardvark_amd.h): expanded by the library's host
function avk_results_expand it must give exactly the wide arrays — through both packers of the kernel-logic emulator here, on the kernels in the gpu tests."""
import ctypes as C

import numpy as np
import pytest

import aardvark_amd
import emu_lib
import oracle_lib
import scenarios
from aardvark_amd import CompareConfig, synth
from aardvark_amd._abi import ResultBatch
from aardvark_amd.api import group_metrics_from_compact

WIDE = ["status", "ed_h1", "ed_h2", "n_optima", "type_present", "var_expected", "var_observed", "var_class", "var_zyg"]


def cases():
    yield scenarios.golden()
    yield scenarios.fuzz_regions(91, 120, max_vars=6, max_len=10)
    yield scenarios.fuzz_regions(92, 120, max_vars=3, repeat_unit=b"CA")
    yield scenarios.invalid_regions()
    c = scenarios.optimizer_golden_regions()
    yield c[0], c[1]
    contig, batch = synth.config_indel_mix_v2(n_truth=500, contig_len=300_000)
    yield [contig], batch


def check(lib, batch, got, want, owned=None):
    """got holds both forms: the packed form expanded == the wide form of the same call == the oracle"""
    wide = got.expanded(lib, batch)
    for f in WIDE:
        a, b, c = getattr(wide, f), getattr(got, f), getattr(want, f)
        if owned is not None and f.startswith("var_"):
            a, b, c = a[owned], b[owned], c[owned]
        assert np.array_equal(a, b), f
        assert np.array_equal(a, c), f


@pytest.mark.parametrize("exact_shortcut", [False, True])
def test_packed_form_expands_to_the_wide_arrays_kernel_logic(oracle, exact_shortcut):
    """(the hidden exact shortcut gives its regions no map entries for absent types: bit 7 of the region word)"""
    lib = aardvark_amd.load_library()
    emu = emu_lib.load()
    emu.emu_set_device_pack.argtypes = [C.c_int]
    seen_filtered = set()
    for devpack in (0, 2):
        emu.emu_set_device_pack(devpack)
        try:
            for contigs, batch in cases():
                want = oracle_lib.compare_batch(oracle, batch, contigs, threads=4, exact_shortcut=exact_shortcut)
                got = emu_lib.compare_batch(batch, contigs, threads=8, exact_shortcut=exact_shortcut, packed=True)
                check(lib, batch, got, want)
                only = emu_lib.compare_batch(batch, contigs, threads=8, exact_shortcut=exact_shortcut, packed="only", group_metrics=False, bp_groups=True)
                assert only.status is None and np.array_equal(only.region_packed, got.region_packed) and np.array_equal(only.var_packed, got.var_packed)
                ok = want.status == 0
                if not exact_shortcut:  # (the shortcut's regions have no RECORD_BP: the compact groups are not defined for it)
                    assert np.array_equal(group_metrics_from_compact(batch, only)[ok], want.group_metrics[ok])  # the host expansion of the BASEPAIR groups reads the packed bytes
                seen_filtered |= set(((got.region_packed[:batch.n_regions] >> np.uint64(7)) & np.uint64(1)).tolist())
        finally:
            emu.emu_set_device_pack(0)
    assert seen_filtered == {0, 1}  # (failed regions have none either)


def test_accessors_of_the_header_agree_with_the_expansion():
    """avk_rp_* / avk_vp_* compiled as C, on words made by avk_rp_make / avk_vp_make: saturation of the distances, the class rule of variant_metrics.rs:43-101"""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = r'''
    #include <stdio.h>
    #include "aardvark_amd.h"
    int main(void) {
      uint64_t w = avk_rp_make(21, 5, 1u << 20, 70000, 0x00ff);
      printf("%d %u %u %u %u\n", avk_rp_status(w), avk_rp_ed_h1(w), avk_rp_ed_h2(w), avk_rp_n_optima(w), avk_rp_filtered_types(w));
      w = avk_rp_make(0, 1048574, 7, 3, 1u << AVK_VT_SNV);
      printf("%d %u %u %u %u %u\n", avk_rp_status(w), avk_rp_ed_h1(w), avk_rp_ed_h2(w), avk_rp_n_optima(w), avk_rp_filtered_types(w), (unsigned)avk_rp_type_present(w, 1u << AVK_VT_SNV));
      w = avk_rp_make(0, 0, 0, 1, AVK_FILTERED_TYPE_MASK | 1u << AVK_VT_UNKNOWN);
      printf("%u %u\n", avk_rp_filtered_types(w), (unsigned)avk_rp_type_present(w, 1u << AVK_VT_UNKNOWN));
      for (unsigned ea = 0; ea < 3; ++ea) for (unsigned oa = 0; oa < 3; ++oa) {
        uint8_t b = avk_vp_make(ea, oa, AVK_ZYG_HOM_ALT);
        printf("%u%u%u%u%u ", avk_vp_expected(b), avk_vp_observed(b), avk_vp_zyg(b), avk_vp_class(b, 0), avk_vp_class(b, 1));
      }
      printf("\n");
      return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(probe)
        subprocess.check_call(["gcc", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-o", os.path.join(d, "p"), os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).decode().split("\n")
    assert out[0].split() == ["21", "5", "1048575", str(70000 & 0xFFFF), "0"]
    assert out[1].split() == ["0", "1048574", "7", "3", "0", "1"]
    filtered = sum(1 << t for t in (0, 1, 2, 3, 4, 5, 9, 10))
    assert out[2].split() == ["1", str(filtered | 1 << 11)]
    tp, fn, fp = 1, 2, 3
    want = []
    for ea in range(3):
        for oa in range(3):
            ct, cq = (0, 0) if ea == 0 and oa == 0 else ((tp, tp) if ea == oa else (fn, fp))
            want.append("%d%d5%d%d" % (ea, oa, ct, cq))
    assert out[3].split() == want


def test_expansion_checks_its_arguments():
    lib = aardvark_amd.load_library()
    contigs, batch = scenarios.golden()
    wide_only = ResultBatch(batch, group_metrics=False)
    with pytest.raises(ValueError):
        wide_only.expanded(lib, batch)  # nothing packed to expand


@pytest.mark.gpu
@pytest.mark.parametrize("device_pack", [1, 0])
def test_packed_form_on_the_gpu(oracle, device_pack):
    ctx = aardvark_amd.Context(0)
    try:
        ctx.set_option("device_pack", device_pack)
        ctx.set_option("lane_min_regions", 0)
        ctx.set_option("lane_min_batch", 0)
        for contigs, batch in cases():
            want = oracle_lib.compare_batch(oracle, batch, contigs, threads=8)
            ctx.upload_reference(contigs)
            got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, packed=True)
            check(ctx.lib, batch, got, want)
            # the packed form alone, into pinned arrays, from a resident batch
            res = ctx.pinned_results(batch, packed="only")
            rb = ctx.upload(batch)
            ctx.compare_resident(rb)
            ro = res.c_struct()
            ctx._check(ctx.lib.avk_results_download(ctx.handle, rb.handle, C.byref(ro)))
            assert np.array_equal(res.region_packed, got.region_packed) and np.array_equal(res.var_packed, got.var_packed)
            rb.free()
        # a contig at the benchmark's density
        contig, batch = synth.config_indel_mix_v2(n_truth=60_000, contig_len=24_000_000)
        ctx.upload_reference([contig])
        want = oracle_lib.compare_batch(oracle, batch, [contig], threads=8)
        got = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, packed=True)
        check(ctx.lib, batch, got, want)
        only = ctx.solve_compare_regions(batch, CompareConfig(enable_sequences=False), group_metrics=False, packed="only")
        assert np.array_equal(only.region_packed, got.region_packed) and np.array_equal(only.var_packed, got.var_packed)
        assert np.array_equal(only.tally, want.tally)
    finally:
        ctx.close()
