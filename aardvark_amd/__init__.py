"""aardvark_amd — MI355X-native (gfx950) solver for the per-region hot path of `aardvark compare`.

The product is `libaardvark_amd.so` (hand-written HIP kernels behind the C-ABI of
include/aardvark_amd.h).  This package is the thin Python plumbing around it: ctypes bindings,
a host-side mirror of the reference's `solve_compare_region` interface, and the synthetic
workload generator used by the benchmark.  There is no CPU path: if the shared library is
missing or no HIP device is present, loading / context creation raises.
"""
import os as _os

# The solver launches on six HIP streams side by side (wave-per-region launches, three lane classes, two solo launches).  The HIP runtime
# maps streams onto 4 hardware queues unless told otherwise, and streams that share a queue run one after the other: with 8 queues a
# whole-genome step takes 5.0 ms instead of 7.0 (round 2), and with a communicator (RCCL) in the process 8 are not enough either (round 3: 6.0 ms
# against 3.8 with 16 or 24: profiles/r03_rccl_queues.txt).  The runtime reads the variable when it initialises, so this only
# helps when nothing in the process has used HIP yet; the caller's own setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from ._abi import (CLASSES, FIELDS, N_FIELDS, N_GROUPS, ST_NAMES, TALLY_LEN, VARIANT_TYPES, ZYGOSITIES, CompactBatch, PackedBatch, RegionBatch,
                   ResultBatch)
from .api import AardvarkAmdError, CompareConfig, Context, library_path, load_library

__all__ = ["Context", "CompareConfig", "RegionBatch", "CompactBatch", "PackedBatch", "ResultBatch", "AardvarkAmdError", "load_library", "library_path",
           "VARIANT_TYPES", "ZYGOSITIES", "CLASSES", "FIELDS", "N_GROUPS", "N_FIELDS", "TALLY_LEN", "ST_NAMES"]
