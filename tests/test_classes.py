"""Comparator classes: machines built with another comparator than
ACM_CMP_DEFAULT over 1- or 2-byte symbols, flattened over the comparator's symbol classes.
CPU side here (class enumeration, tables, blob); the device side is in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from aho_corasick_1975_amd import binding
from oracle import pyoracle as po
from tests import flatwalk


def fn_ptr(kat, name):
    return C.cast(getattr(kat, name), C.c_void_p)


def pair(kat, cmp_name, sym, keywords):
    """The product machine and the oracle machine with the same C comparator and keywords."""
    m = acm.Machine(sym, cmp=fn_ptr(kat, cmp_name))
    o = po.Oracle(sym, po.MEYER85, cmp=fn_ptr(kat, cmp_name))
    for kw in keywords:
        m.add_keyword(kw)
        o.add_keyword(kw)
    return m, o


def test_case_insensitive_bytes_classes(kat, novel_bytes):
    """The byte analogue of the reference's alphacmp run (generic_test.c:48-54): he/she/his/hers
    entered in mixed case, matched case-insensitively over the novel."""
    m, o = pair(kat, "kat_casecmp8", 1, [b"He", b"SHE", b"his", b"hErs"])
    flat = m.flatten_classes()
    assert flat.n_classes == 256 - 26
    cm = flat.class_map
    assert all(cm[ord(c)] == cm[ord(c.upper())] for c in "abcxyz") and cm[ord("a")] != cm[ord("b")]
    assert np.all(np.diff(cm[np.argsort(cm, kind="stable")].astype(int)) >= 0)
    want = o.scan(novel_bytes)
    text_classes = cm[np.frombuffer(novel_bytes, np.uint8)].astype(np.uint8)
    got = flatwalk.walk_csr(flat, text_classes)
    assert got.size == want.size and np.array_equal(got, want)
    assert np.array_equal(flatwalk.walk_dense(flat, text_classes), want)
    lower = np.frombuffer(novel_bytes.lower(), np.uint8)
    m2, o2 = pair(kat, "kat_casecmp8", 1, [b"he", b"she", b"his", b"hers"])
    assert want.size == o2.scan(lower.tobytes()).size > 11676       # more than the case-sensitive run
    # the dictionary's own spelling comes back, edge by edge the FIRST letter inserted there
    # (reference: letters[] point at the machine's stored letters, aho_corasick.c:477-479 -- "his"
    # and "hErs" run over the 'H' and 'e' that "He" put in)
    assert [flat.keyword(k).tobytes() for k in range(4)] == [b"He", b"SHE", b"His", b"Hers"]


def test_interleaved_classes_mod7(kat):
    rng = np.random.default_rng(5)
    kws = [bytes(rng.integers(0, 256, size=rng.integers(1, 6)).astype(np.uint8)) for _ in range(60)]
    m, o = pair(kat, "kat_mod7cmp8", 1, kws)
    flat = m.flatten_classes()
    assert flat.n_classes == 7 and np.array_equal(flat.class_map, np.arange(256) % 7)
    assert flat.info.n_keywords == o.nb_keywords and flat.info.n_states == o.nb_states
    text = rng.integers(0, 256, size=20000).astype(np.uint8)
    want = o.scan(text)
    assert want.size > 100
    assert np.array_equal(flatwalk.walk_csr(flat, flat.class_map[text].astype(np.uint8)), want)
    assert np.array_equal(flatwalk.walk_dense(flat, flat.class_map[text].astype(np.uint8)), want)


def test_u16_case_insensitive_classes(kat):
    kws = [np.array([ord(c) for c in w], np.uint16) for w in ("Été", "STRASSE", "naïve", "Ωmega", "x")]
    m, o = pair(kat, "kat_casecmp16", 2, kws)
    flat = m.flatten_classes()
    cm = flat.class_map
    assert cm.size == 65536 and cm[ord("É")] == cm[ord("é")] and cm[ord("Ω")] == cm[ord("ω")] and cm[ord("s")] != cm[ord("t")]
    text = np.array([ord(c) for c in "l'été ÉTÉ Strasse straße NAÏVE ωMEGA ΩMEGa xX " * 40], np.uint16)
    want = o.scan(text)
    assert want.size >= 40 * 8
    assert np.array_equal(flatwalk.walk_csr(flat, cm[text]), want)
    assert flat.keyword(0).tolist() == [ord(c) for c in "Été"]


def test_u32_case_insensitive_classes_of_the_dictionary_symbols(kat):
    """4-byte symbols (the reference's wchar_t + alphacmp, generic_test.c:48-54,73-99): the classes
    are those of the dictionary's own symbols, 1 .. n in comparator order; a text is mapped symbol
    by symbol with the comparator (here on the CPU, as the plan does for symbols it has not met)."""
    kat.setlocale_utf8()
    words = ["he", "She", "SHEERS", "his", "hi", "Hers", "ushers", "abcde", "bcd", "hers", "Été", "étÉ"]
    kws = [np.array([ord(c) for c in w], np.uint32) for w in words]
    m, o = pair(kat, "kat_casecmp32", 4, kws)
    flat = m.flatten_classes()
    assert flat.info.sym_bytes == 4 and flat.info.n_keywords == o.nb_keywords == 10 and flat.info.n_states == o.nb_states
    keys, cls, reps = flat.keys32, flat.keys32_class, flat.class_rep32
    assert np.all(np.diff(keys.astype(np.int64)) > 0) and cls.min() == 1 and cls.max() == flat.n_classes == reps.size
    k2c = dict(zip(keys.tolist(), cls.tolist()))
    assert k2c[ord("S")] == k2c[ord("s")] and k2c[ord("É")] == k2c[ord("é")] and k2c[ord("h")] != k2c[ord("i")]
    folded = [chr(r).lower() for r in reps.tolist()]
    assert folded == sorted(folded, key=ord)                 # comparator order = order of the lower-case code points
    fold2c = {chr(k).lower(): c for k, c in k2c.items()}
    text = "He found his pencil, but SHE could not find hErs; ÉTÉ, sheers! USHERS abcdE" * 30
    classes = np.array([fold2c.get(ch.lower(), 0) for ch in text], np.uint32)
    want = o.scan(np.array([ord(c) for c in text], np.uint32))
    assert want.size > 300
    assert np.array_equal(flatwalk.walk_csr(flat, classes), want)
    assert flat.keyword(1).tolist() == [ord(c) for c in "She"]      # the dictionary's own spelling
    with pytest.raises(binding.ACMError):
        flat.to_bytes()                                      # no serialised form without the comparator
    # a comparator that is no order over the dictionary's symbols is refused
    bad = acm.Machine(4, cmp=fn_ptr(kat, "kat_cyclic_cmp32"))
    for w in ("abc", "bca"):
        bad.add_keyword(np.array([ord(c) for c in w], np.uint32))
    with pytest.raises(binding.ACMError):
        bad.flatten_classes()


def test_default_comparator_classes_are_identity(kat):
    m = acm.Machine(1)
    for kw in (b"he", b"she", b"his", b"hers"):
        m.add_keyword(kw)
    flat = m.flatten_classes()
    assert flat.n_classes == 256 and np.array_equal(flat.class_map, np.arange(256))
    plain = m.flatten()
    for k in ("row_ptr", "edge_sym", "edge_next", "fail", "nb_outputs", "term_kw"):
        assert np.array_equal(getattr(flat, k), getattr(plain, k))


def test_inconsistent_comparator_is_refused(kat):
    m = acm.Machine(1, cmp=fn_ptr(kat, "kat_cyclic_cmp8"))
    with pytest.raises(binding.ACMError) as ei:
        m.flatten_classes()
    assert ei.value.code == binding.ACM_GPU_E_INELIGIBLE
    # and the plain entry points refuse any custom comparator, as before
    m2 = acm.Machine(1, cmp=fn_ptr(kat, "kat_casecmp8"))
    with pytest.raises(binding.ACMError):
        m2.flatten()
    # 8-byte symbols have no class path (4-byte ones do: classes of the dictionary's own symbols)
    m8 = acm.Machine(8, cmp=fn_ptr(kat, "kat_alphacmp"))
    with pytest.raises(binding.ACMError):
        m8.flatten_classes()


def test_class_machine_blob_round_trip(kat, tmp_path):
    m, o = pair(kat, "kat_casecmp8", 1, [b"He", b"SHE", b"his", b"hErs", b"Q"])
    flat = m.flatten_classes()
    blob = flat.to_bytes()
    back = binding.FlatTables.from_bytes(blob)
    assert back.n_classes == flat.n_classes and np.array_equal(back.class_map, flat.class_map)
    assert np.array_equal(back.edge_letter, flat.edge_letter) and np.array_equal(back.edge_sym, flat.edge_sym)
    assert [back.keyword(k).tobytes() for k in range(5)] == [b"He", b"SHE", b"His", b"Hers", b"Q"]
    # an edge whose letter is not in the class it is filed under is caught
    words = np.frombuffer(blob[80:], dtype="<u4").copy()
    words[-1] = ord("z")                     # last edge_letter
    from tests.test_flat_blob import refnv
    with pytest.raises(RuntimeError):
        binding.FlatTables.from_bytes(refnv(blob[:80] + words.tobytes()))
