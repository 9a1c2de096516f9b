"""tools/cpu_loop.c (bench.py's second CPU leg): the reference's caller loop over the drop-in's own
per-symbol API, several threads on one shared machine, equals the oracle on the same text."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from oracle import pyoracle as po
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def cpuloop():
    so = os.path.join(ROOT, "tools", "libcpuloop.so")
    src = os.path.join(ROOT, "tools", "cpu_loop.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), src, "-o", so],
                       check=True)
    C.CDLL(acm.binding.library_path(), mode=C.RTLD_GLOBAL)
    L = C.CDLL(so)
    L.acm_cpu_loop.restype = C.c_uint64
    L.acm_cpu_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
    return L


@pytest.mark.parametrize("sym,K", [(1, 1000), (4, 2000)])
def test_dropin_caller_loop_equals_oracle(cpuloop, sym, K):
    kd, ko = acm.synth.keywords(K, sym_bytes=sym, vocab=500)
    m = acm.Machine(sym)
    m.add_keywords_packed(kd, ko, ids_as_values=True)
    o = po.Oracle(sym)
    o.add_keywords_packed(kd, ko)
    text = acm.synth.text(1 << 20, kd, ko, sym_bytes=sym, vocab=500)
    want = o.scan_mt(text, 1)
    assert want[0] > 200
    for threads in (1, 3, 8):
        d = C.c_uint64(0)
        n = cpuloop.acm_cpu_loop(m.handle, text.ctypes.data, text.size, sym, m.lmax, threads, C.byref(d))
        assert (int(n), int(d.value)) == want
