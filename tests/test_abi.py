"""The C-ABI library loads and exports every symbol include/aardvark_amd.h declares; the ctypes
mirrors have the C layouts.  No compute calls (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

import aardvark_amd
from aardvark_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aardvark_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = aardvark_amd.load_library()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libaardvark_amd.so does not export %s" % n
    assert b"gfx950" in lib.avk_version()


def test_struct_layouts_match_the_header():
    probe = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "aardvark_amd.h"
    int main(void) {
      printf("%zu %zu %zu\n", sizeof(avk_region_batch), sizeof(avk_compare_config), sizeof(avk_result_batch));
      printf("%zu %zu %zu\n", offsetof(avk_region_batch, n_variants), offsetof(avk_region_batch, allele_bytes_len), offsetof(avk_result_batch, tally));
      printf("%d %d %d\n", AVK_N_GROUPS, AVK_N_FIELDS, AVK_TALLY_LEN);
      return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(probe)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "p"), os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).decode().split("\n")
    sizes = [int(x) for x in out[0].split()]
    assert sizes == [C.sizeof(_abi.AvkRegionBatch), C.sizeof(_abi.AvkCompareConfig), C.sizeof(_abi.AvkResultBatch)]
    offs = [int(x) for x in out[1].split()]
    assert offs == [_abi.AvkRegionBatch.n_variants.offset, _abi.AvkRegionBatch.allele_bytes_len.offset, _abi.AvkResultBatch.tally.offset]
    assert [int(x) for x in out[2].split()] == [_abi.N_GROUPS, _abi.N_FIELDS, _abi.TALLY_LEN]


def test_context_creation_fails_loudly_without_a_gpu():
    """There is no CPU path behind the C-ABI: without a HIP device the library refuses to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(aardvark_amd.AardvarkAmdError):
        aardvark_amd.Context(0)


def test_product_library_does_not_link_the_oracle():
    out = subprocess.check_output(["ldd", aardvark_amd.library_path()]).decode()
    assert "oracle" not in out and "avk_emu" not in out
    syms = subprocess.check_output(["nm", "-D", aardvark_amd.library_path()]).decode()
    assert "orc_" not in syms and "emu_compare_batch" not in syms
