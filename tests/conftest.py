import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
HELPERS = os.path.join(ROOT, "tests", "helpers")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_kat_helpers():
    so = os.path.join(HELPERS, "libkat.so")
    src = os.path.join(HELPERS, "kat_helpers.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", so, src], check=True)
    return so


@pytest.fixture(scope="session")
def kat():
    import ctypes as C
    L = C.CDLL(_build_kat_helpers())
    L.kat_rand_letters.argtypes = [C.c_void_p, C.c_size_t]
    L.kat_srand.argtypes = [C.c_uint]
    L.kat_read_novel.restype = C.c_long
    L.kat_read_novel.argtypes = [C.c_char_p, C.c_void_p, C.c_long]
    L.setlocale_utf8 = L.kat_setlocale_utf8
    return L


@pytest.fixture(scope="session")
def novel_bytes():
    with open(os.path.join(GOLDEN, "mrs_dalloway.txt"), "rb") as f:
        return f.read()
