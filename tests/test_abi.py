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
    inline = set(re.findall(r"static inline [a-z0-9_]+\s+(avk_[a-z0-9_]+)\s*\(", src))  # the accessors of the packed result form: defined in the header
    assert len(inline) >= 10
    return sorted(set(re.findall(r"\b(avk_[a-z0-9_]+)\s*\(", src)) - inline)


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
      printf("%zu %zu\n", offsetof(avk_result_batch, region_packed), offsetof(avk_result_batch, var_packed));
      printf("%d %d %d\n", AVK_N_GROUPS, AVK_N_FIELDS, AVK_TALLY_LEN);
      printf("%zu %zu %zu %zu\n", sizeof(avk_compact_batch), sizeof(avk_packed_batch), offsetof(avk_packed_batch, n_variants), offsetof(avk_packed_batch, allele_bytes_len));
      printf("%zu %zu %zu %zu %zu\n", sizeof(avk_multi_batch), sizeof(avk_packed_multi_batch), offsetof(avk_packed_multi_batch, in_cnt), offsetof(avk_packed_multi_batch, n_variants),
             offsetof(avk_packed_multi_batch, allele_bytes_len));
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
    assert [int(x) for x in out[2].split()] == [_abi.AvkResultBatch.region_packed.offset, _abi.AvkResultBatch.var_packed.offset]
    out = out[:2] + out[3:]
    assert [int(x) for x in out[2].split()] == [_abi.N_GROUPS, _abi.N_FIELDS, _abi.TALLY_LEN]
    assert [int(x) for x in out[3].split()] == [C.sizeof(_abi.AvkCompactBatch), C.sizeof(_abi.AvkPackedBatch), _abi.AvkPackedBatch.n_variants.offset, _abi.AvkPackedBatch.allele_bytes_len.offset]
    from aardvark_amd import merge
    assert [int(x) for x in out[4].split()] == [C.sizeof(merge.AvkMultiBatch), C.sizeof(merge.AvkPackedMultiBatch), merge.AvkPackedMultiBatch.in_cnt.offset,
                                                merge.AvkPackedMultiBatch.n_variants.offset, merge.AvkPackedMultiBatch.allele_bytes_len.offset]


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


def test_packer_edit_distance_matches_the_oracle():
    """Variant::alt_ed is computed by the batch packer on the host (no GPU): fuzz it against the oracle's wfa_ed."""
    import random
    import oracle_lib
    lib = aardvark_amd.load_library()
    orc = oracle_lib.load()
    rng = random.Random(7)
    cases = [(b"A", b"A"), (b"A", b"C"), (b"A", b"ACGT"), (b"ACGT", b"A"), (b"ACGT", b"T"), (b"AAAA", b"AA"), (b"ACAC", b"CACA")]
    for _ in range(3000):
        n = rng.choice([1, 1, 2, 3, 5, 8, 20, 60])
        a = bytes(rng.choice(b"ACGT") for _ in range(n))
        kind = rng.random()
        if kind < 0.3:
            b = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([1, 2, 3, 7, 30])))
        else:  # a mutated copy
            b = bytearray(a)
            for _ in range(rng.randint(0, 4)):
                op = rng.random()
                pos = rng.randrange(len(b) + 1)
                if op < 0.34 and len(b) > 1:
                    del b[min(pos, len(b) - 1)]
                elif op < 0.67:
                    b.insert(pos, rng.choice(b"ACGT"))
                elif len(b):
                    b[min(pos, len(b) - 1)] = rng.choice(b"ACGT")
            b = bytes(b) or b"A"
        cases.append((a, b))
    for a, b in cases:
        pa, la = oracle_lib.b2p(a)
        pb, lb = oracle_lib.b2p(b)
        want = orc.orc_wfa_ed(pa, la, pb, lb)
        assert lib.avk_edit_distance(a, len(a), b, len(b)) == want, (a, b)


def test_pinned_arrays_outlive_close_without_touching_freed_memory():
    """Context.host_array blocks are freed with the last array that views them, the native context with the last block after close()
    (the counting is host logic: a stand-in library records the calls, no GPU needed)"""
    import gc
    import ctypes as C
    import numpy as np
    from aardvark_amd import api

    class FakeLib:
        def __init__(self):
            self.freed, self.destroyed = [], 0
            self.blocks = {}

        def avk_host_alloc(self, handle, nbytes):
            buf = (C.c_uint8 * nbytes)()
            self.blocks[C.addressof(buf)] = buf
            return C.addressof(buf)

        def avk_host_free(self, handle, p):
            self.freed.append(p)

        def avk_ctx_destroy(self, handle):
            self.destroyed += 1

    lib = FakeLib()
    ctx = api.Context.__new__(api.Context)
    ctx.lib, ctx.handle, ctx._contigs = lib, C.c_void_p(1234), None
    ctx._core = api._ContextCore(lib, ctx.handle)
    a = ctx.host_array((16,), np.uint32)
    b = ctx.host_array((4, 4), np.uint8)
    view = a[2:5]
    a[:] = 7
    ctx.close()
    assert lib.destroyed == 0 and lib.freed == []  # arrays are alive: nothing is freed yet
    assert int(view.sum()) == 21
    del a
    gc.collect()
    assert lib.freed == []  # `view` still keeps the first block
    del b
    gc.collect()
    assert len(lib.freed) == 1 and lib.destroyed == 0
    del view
    gc.collect()
    assert len(lib.freed) == 2 and lib.destroyed == 1
