"""MI355X-native Aho-Corasick scan engine behind the acm_* C API of farhiongit/aho-corasick-1975.

The product is the C-ABI shared library `libac75_amd.so` (host acm_* API in C, flattener in C,
scan kernels in HIP for gfx950; sources under csrc/, headers under /include).  This package is
the thin Python host layer over that C ABI: ctypes bindings that mirror the reference's API
names, plus torch plumbing (device buffers, streams, torch.distributed) for the bulk scan.

There is no CPU scan path here: the bulk scan raises if the library or a GPU is missing.
"""
from .binding import (  # noqa: F401
    ACMError, Machine, Plan, Stream, MultiScan, Comm, FlatTables, RECORD_DTYPE, build_native, lib, library_path,
)
from . import synth  # noqa: F401
from . import sharded  # noqa: F401
