"""aardvark_amd — MI355X-native (gfx950) solver for the per-region hot path of `aardvark compare`.

The product is `libaardvark_amd.so` (hand-written HIP kernels behind the C-ABI of
include/aardvark_amd.h).  This package is the thin Python plumbing around it: ctypes bindings,
a host-side mirror of the reference's `solve_compare_region` interface, and the synthetic
workload generator used by the benchmark.  There is no CPU path: if the shared library is
missing or no HIP device is present, loading / context creation raises.
"""
from ._abi import (CLASSES, FIELDS, N_FIELDS, N_GROUPS, ST_NAMES, TALLY_LEN, VARIANT_TYPES, ZYGOSITIES, RegionBatch,
                   ResultBatch)
from .api import AardvarkAmdError, CompareConfig, Context, library_path, load_library

__all__ = ["Context", "CompareConfig", "RegionBatch", "ResultBatch", "AardvarkAmdError", "load_library", "library_path",
           "VARIANT_TYPES", "ZYGOSITIES", "CLASSES", "FIELDS", "N_GROUPS", "N_FIELDS", "TALLY_LEN", "ST_NAMES"]
