"""acm_scan on every machine the reference's API can make ("eligible for the GPU only
when cmp == ACM_CMP_DEFAULT and the symbol size is 1, 2, 4 or 8; otherwise it runs loop a9 on the
CPU"): which path runs, and that what the host loop returns is the caller loop's record set.  The
host loop is the PRODUCT's own acm_match / acm_get_match steps (acm_host.c), checked here against the
definition-level brute force and the oracle; a missing GPU stays an error for machines the GPU can
take (no silent fallback)."""
import ctypes as C

import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from aho_corasick_1975_amd import binding
from oracle import pyoracle as po
from tests.brute import brute_records

PATH_GPU, PATH_CLASSES, PATH_LOOP = 1, 2, 3


def _raw_machine(sym_bytes, keywords):
    """ACM_CMP_DEFAULT over symbols of `sym_bytes` bytes (any size: memcmp), keywords = lists of byte strings"""
    L = acm.lib()
    arg = C.c_size_t(sym_bytes)
    keep = [arg]
    h = L.acm_create(C.c_void_p.in_dll(L, "ACM_CMP_DEFAULT"), C.cast(C.pointer(arg), C.c_void_p), None)
    for kw in keywords:
        buf = np.frombuffer(b"".join(kw), dtype=np.uint8).copy()
        keep.append(buf)
        cur = C.c_void_p(L.acm_initiate(h))
        for i in range(len(kw)):
            L.acm_insert_letter_of_keyword(C.byref(cur), buf.ctypes.data + i * sym_bytes)
        L.acm_insert_end_of_keyword(C.byref(cur), None, None)
    return h, keep


def _scan(h, text_bytes, n_symbols, cap=4096):
    L = acm.lib()
    t = np.frombuffer(text_bytes, dtype=np.uint8).copy()
    out = np.zeros(cap, dtype=binding.RECORD_DTYPE)
    n = C.c_uint64(0)
    rc = L.acm_scan(h, t.ctypes.data, n_symbols, out.ctypes.data, cap, C.byref(n))
    return rc, out[:min(int(n.value), cap)], int(n.value)


def test_three_byte_symbols_run_the_caller_loop_on_the_host():
    rng = np.random.default_rng(3)
    alphabet = [bytes(rng.integers(0, 256, size=3).astype(np.uint8)) for _ in range(5)]
    kws = [[alphabet[int(i)] for i in rng.integers(0, 5, size=int(rng.integers(1, 5)))] for _ in range(40)]
    text = [alphabet[int(i)] for i in rng.integers(0, 5, size=3000)]
    h, keep = _raw_machine(3, kws)
    L = acm.lib()
    assert L.acm_scan_path(h) == 0
    rc, got, total = _scan(h, b"".join(text), len(text), cap=20000)
    assert rc == 0 and L.acm_scan_path(h) == PATH_LOOP
    want = brute_records(kws, text)
    assert total == want.size and np.array_equal(got, want)
    # a buffer that is too small: the total is reported, the error says so, what fits is the head of the loop's order
    rc, head, total2 = _scan(h, b"".join(text), len(text), cap=7)
    assert rc == binding.ACM_GPU_E_OVERFLOW and total2 == want.size and np.array_equal(head, want[:7])
    L.acm_release(h)


def test_comparator_without_a_declared_symbol_size_is_ineligible_and_the_declaration_is_checked(kat):
    m = acm.Machine(1, cmp=C.cast(kat.kat_cyclic_cmp8, C.c_void_p))
    m.add_keyword(b"ab")
    n = C.c_uint64(0)
    out = np.zeros(8, dtype=binding.RECORD_DTYPE)
    t = np.frombuffer(b"xxabxx", np.uint8).copy()
    assert acm.lib().acm_scan(m.handle, t.ctypes.data, t.size, out.ctypes.data, 8, C.byref(n)) == binding.ACM_GPU_E_INELIGIBLE
    assert acm.lib().acm_set_symbol_bytes(m.handle, 0) == binding.ACM_GPU_E_ARG
    plain = acm.Machine(2)
    assert acm.lib().acm_set_symbol_bytes(plain.handle, 4) == binding.ACM_GPU_E_ARG      # ACM_CMP_DEFAULT says 2 itself
    assert acm.lib().acm_set_symbol_bytes(plain.handle, 2) == 0


def test_inconsistent_comparator_runs_the_caller_loop_on_the_host(kat):
    """kat_cyclic_cmp8 is no order (a < b < c < a): acm_flatten_classes refuses it, so no GPU plan can
    exist for the machine -- acm_scan runs the loop itself, with the machine's own comparator."""
    cmp = C.cast(kat.kat_cyclic_cmp8, C.c_void_p)
    kws = [b"he", b"she", b"his", b"hers", b"e"]
    m = acm.Machine(1, cmp=cmp)
    o = po.Oracle(1, po.MEYER85, cmp=cmp)
    for kw in kws:
        m.add_keyword(kw)
        o.add_keyword(kw)
    m.set_symbol_bytes(1)
    text = b"ushers and heroes: she sells his shells, hers too" * 7
    got = m.scan_host(text)
    assert m.scan_path == PATH_LOOP
    want = o.scan(text)
    assert want.size > 50 and np.array_equal(got, want)


def test_no_silent_fallback_without_a_gpu(kat):
    """A machine the GPU CAN take (a consistent comparator over bytes; ACM_CMP_DEFAULT over bytes)
    must not be scanned on the host when no device is there: the call fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU: the GPU tests cover the machine")
    m = acm.Machine(1, cmp=C.cast(kat.kat_casecmp8, C.c_void_p))
    m.add_keyword(b"He")
    m.set_symbol_bytes(1)
    with pytest.raises(binding.ACMError) as e:
        m.scan_host(b"he said HE would")
    assert e.value.code in (binding.ACM_GPU_E_NODEVICE, binding.ACM_GPU_E_HIP)
    plain = acm.Machine(1)
    plain.add_keyword(b"he")
    with pytest.raises(binding.ACMError) as e:
        plain.scan_host(b"he said he would")
    assert e.value.code in (binding.ACM_GPU_E_NODEVICE, binding.ACM_GPU_E_HIP)
    assert m.scan_path == 0 and plain.scan_path == 0


@pytest.mark.gpu
def test_reference_alphacmp_through_plain_acm_scan_on_the_gpu(kat, novel_bytes):
    """The reference's flagship use -- any comparator (aho_corasick.h:33-45), here its own alphacmp over
    wchar_t (generic_test.c:48-54) and the byte analogue -- through plain acm_scan: the symbol size is
    declared once, the scan runs on the GPU over the comparator's classes."""
    import torch
    assert torch.cuda.is_available()
    for sym, name, dt in ((4, "kat_casecmp32", np.uint32), (1, "kat_casecmp8", np.uint8)):
        cmp = C.cast(getattr(kat, name), C.c_void_p)
        m = acm.Machine(sym, cmp=cmp)
        o = po.Oracle(sym, po.MEYER85, cmp=cmp)
        for kw in (b"He", b"SHE", b"his", b"hErs", b"Mrs", b"dalloway"):
            w = np.frombuffer(kw, np.uint8).astype(dt)
            m.add_keyword(w)
            o.add_keyword(w)
        m.set_symbol_bytes(sym)
        text = np.frombuffer(novel_bytes[:200000], np.uint8).astype(dt)
        got = m.scan_host(text)
        assert m.scan_path == PATH_CLASSES
        want = o.scan(text)
        assert want.size > 5000 and np.array_equal(got, want)
    plain = acm.Machine(1)
    plain.add_keyword(b"he")
    plain.scan_host(novel_bytes[:5000])
    assert plain.scan_path == PATH_GPU
