"""The product's host acm_* API (aho-corasick-1975_amd/csrc/acm_host.c) against the reference's
known answers and against the oracle.  These read like the reference's own example programs
(examples/test.c, examples/aho_corasick_generic_test.c) re-expressed over ctypes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from oracle import pyoracle as po
from tests.brute import brute_records
from tests.test_oracle_kat import (GENERIC_KEYWORDS, GENERIC_SCAN_ORDER, GENERIC_TEXT, README_LINE, README_TEXT)


def test_library_exports_every_declared_symbol():
    L = acm.lib()
    for name in acm.binding.EXPORTS:
        assert getattr(L, name) is not None
    assert C.c_int.in_dll(L, "ACM_INCREMENTAL_STRING_MATCHING").value == 1


def test_export_list_covers_the_headers():
    """Every function the public headers declare is in the binding's EXPORTS list (which the two
    tests around this one check against the library), so the list cannot fall behind the headers."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in ("acm.h", "acm_gpu.h"):
        text = open(os.path.join(root, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                 # comments mention functions too
        declared |= set(re.findall(r"\b(acm_[a-z0-9_]+)\s*\(", text))
        declared |= set(re.findall(r"\b(ACM_[A-Z_]+)\s*;", text))
    missing = sorted(n for n in declared if n not in acm.binding.EXPORTS)
    assert not missing, missing


def test_headers_declare_what_the_library_exports():
    """include/acm.h + include/acm_gpu.h compile as C11 and every declared function links."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "#include \"aho_corasick.h\"\n#include \"acm_gpu.h\"\nint main(void){" + "".join(
        "(void)%s;" % n for n in acm.binding.EXPORTS) + "return 0;}\n"
    exe = os.path.join(root, "tests", "helpers", "link_check")
    p = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-x", "c", "-",
                        "-o", exe, "-L", os.path.dirname(acm.library_path()), "-lac75_amd",
                        "-Wl,-rpath," + os.path.dirname(acm.library_path())], input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    os.remove(exe)


def test_readme_known_answer():
    """examples/test.c -> README.md:93, through the product's per-symbol API."""
    m = acm.Machine(1)
    for w in (b"he", b"she", b"his", b"hers"):
        assert m.add_keyword(w) is None
    assert m.nb_keywords == 4
    by_pos = {}
    for i, length, word, _ in m.match_loop(README_TEXT):
        by_pos.setdefault(i, []).append((length, bytes(word)))
    line = ""
    for i in sorted(by_pos):
        for length, word in reversed(by_pos[i]):     # test.c iterates j = nb..1
            line += " %d:%s" % (i + 2 - length, word.decode())
    assert line == README_LINE


def test_generic_part1_custom_comparator_values_and_order(kat):
    """generic_test.c:62-164 with the case-insensitive wide-char comparator: asserts :70, :114,
    :117; 21 keywords; dictionary spelling in MatchHolder; scan order of SURVEY.md App. C."""
    L = acm.lib()
    m = L.acm_create(C.cast(kat.kat_alphacmp, C.c_void_p), None, None)
    cur = C.c_void_p(L.acm_initiate(m))
    a = C.c_uint32(ord("a"))
    assert L.acm_match(C.byref(cur), C.byref(a)) == 0
    keep = []
    ins = C.c_void_p(L.acm_initiate(m))
    root = ins.value
    for index, (kw, check, total) in enumerate(GENERIC_KEYWORDS):
        letters = np.array([ord(ch) for ch in kw], dtype=np.uint32)
        val = C.c_size_t(index)
        keep += [letters, val]
        for i in range(letters.size):
            L.acm_insert_letter_of_keyword(C.byref(ins), letters.ctypes.data + 4 * i)
        prev = L.acm_insert_end_of_keyword(C.byref(ins), C.addressof(val), None)
        assert ins.value == root                                  # cursor back at the root (:360)
        assert (0 if prev else 1) == check
        if prev:
            pv = C.c_size_t.from_address(prev)
            pv.value += val.value
            assert pv.value == total
        else:
            assert val.value == total
    assert L.acm_nb_keywords(m) == 21
    text = np.array([ord(ch) for ch in GENERIC_TEXT], dtype=np.uint32)
    h = acm.binding.MatchHolder()
    L.acm_matcher_init(C.byref(h))
    cur = C.c_void_p(L.acm_initiate(m))
    seen = []
    for i in range(text.size):
        nb = L.acm_match(C.byref(cur), text.ctypes.data + 4 * i)
        for j in range(nb):
            L.acm_get_match(cur, j, C.byref(h))
            seen.append("".join(chr(C.cast(h.letters[k], C.POINTER(C.c_uint32))[0]) for k in range(h.length)))
        if nb:
            L.acm_get_match(cur, 0, None)                         # NULL matcher is allowed (:467-468)
    L.acm_matcher_release(C.byref(h))
    assert seen == GENERIC_SCAN_ORDER
    # this machine is not eligible for the GPU path: loud error, no fallback
    flat = C.c_void_p()
    assert L.acm_flatten(m, C.byref(flat)) == -1
    L.acm_release(m)


def test_foreach_and_print(kat):
    """acm_foreach_keyword in comparator order and the acm_print drawing (facts recorded in
    SURVEY.md App. C: she(005)[+2] fails to he(002); ushers(018)[+2] -> hers(012);
    abcde(023)[+3] -> bcde(028); uu(038)[+2] -> u(013); 39 states 000-038)."""
    L = acm.lib()
    m = L.acm_create(C.cast(kat.kat_alphacmp, C.c_void_p), None, None)
    keep = []
    ins = C.c_void_p(L.acm_initiate(m))
    for kw, _, _ in GENERIC_KEYWORDS:
        letters = np.array([ord(ch) for ch in kw], dtype=np.uint32)
        keep.append(letters)
        for i in range(letters.size):
            L.acm_insert_letter_of_keyword(C.byref(ins), letters.ctypes.data + 4 * i)
        L.acm_insert_end_of_keyword(C.byref(ins), None, None)
    words = []
    OP = C.CFUNCTYPE(None, acm.binding.MatchHolder)

    def op(holder):
        words.append("".join(chr(C.cast(holder.letters[k], C.POINTER(C.c_uint32))[0]) for k in range(holder.length)))
    cb = OP(op)
    L.acm_foreach_keyword(m, cb)
    assert words == sorted(set(k for k, _, _ in GENERIC_KEYWORDS))
    L.acm_foreach_keyword(m, None)                                # no-op (:524-525)

    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.fputc.argtypes = [C.c_int, C.c_void_p]
    path = os.path.join(os.path.dirname(__file__), "helpers", "print_out.txt")
    f = libc.fopen(path.encode(), b"w")
    PR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

    def pr(stream, letter):
        libc.fputc(C.cast(letter, C.POINTER(C.c_uint32))[0], stream)
        return 1
    prcb = PR(pr)
    L.acm_print(m, f, prcb)
    L.acm_print(m, None, prcb)                                    # NULL stream prints nothing (:589)
    libc.fclose(f)
    drawing = open(path).read()
    os.remove(path)
    assert drawing.startswith("\n(000)---") and drawing.endswith("\n")
    for fact in ("--e-->(005)[+2](v 002)", "--s-->(018)[+2](v 012)", "--e-->(023)[+3](v 028)", "--u-->(038)[+2](v 013)"):
        assert fact in drawing, fact
    assert "(038)" in drawing and "(039)" not in drawing
    L.acm_release(m)


def test_letter_and_value_destructors():
    """Ownership rules of SURVEY.md 8b: a letter whose edge already exists is destroyed at once
    (:306-307), stored letters at release (:111-112); the first non-NULL value is kept with its
    dtor (:358-359) and destroyed at release (:124-125); a later value is NOT taken over."""
    L = acm.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    freed = []
    DT = C.CFUNCTYPE(None, C.c_void_p)

    def dtor(p):
        freed.append(p)
    cb = DT(dtor)
    arg = C.c_size_t(1)
    m = L.acm_create(C.c_void_p.in_dll(L, "ACM_CMP_DEFAULT"), C.cast(C.pointer(arg), C.c_void_p), C.cast(cb, C.c_void_p))
    allocated = []

    def insert(word, value):
        cur = C.c_void_p(L.acm_initiate(m))
        for ch in word:
            p = libc.malloc(1)
            C.c_ubyte.from_address(p).value = ch
            allocated.append(p)
            L.acm_insert_letter_of_keyword(C.byref(cur), p)
        return L.acm_insert_end_of_keyword(C.byref(cur), value, C.cast(cb, C.c_void_p))
    v1, v2 = libc.malloc(8), libc.malloc(8)
    assert insert(b"abc", None) is None
    assert freed == []
    assert insert(b"abd", v1) is None                 # 'a','b' edges exist: those two letters die now
    assert freed == allocated[3:5]
    assert insert(b"abd", v2) == v1                   # previous value returned, v2 not taken over
    assert v2 not in freed
    assert insert(b"abc", v2) is None                 # abc had no value yet: v2 is taken now
    n_before = len(freed)
    L.acm_release(m)
    released = freed[n_before:]
    assert v1 in released and v2 in released
    stored = [allocated[0], allocated[1], allocated[2], allocated[5]]
    assert all(p in released for p in stored)
    assert sorted(freed) == sorted(allocated + [v1, v2])   # everything exactly once


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_interleaved_insert_and_match_equals_oracle(seed):
    """Meyer-85 maintenance in the product vs the oracle's AC-75 rebuild: identical match sets
    after every batch of insertions, cursor carried across insertions (generic_test.c:198-229)."""
    rng = np.random.default_rng(seed)
    m = acm.Machine(1)
    o = po.Oracle(1, po.AC75)
    words = []
    text = bytes(rng.integers(97, 100, size=1500, dtype=np.uint8))
    for k in range(200):
        w = bytes(rng.integers(97, 100, size=int(rng.integers(1, 8)), dtype=np.uint8))
        words.append(w)
        m.add_keyword(w)
        o.add_keyword(w)
        if k % 20 == 19:
            got = [(i, length) for i, length, _, _ in m.match_loop(text)]
            want = o.scan(text)
            assert got == list(zip(want["end_pos"].tolist(), want["length"].tolist()))
    assert m.nb_keywords == o.nb_keywords
    b = brute_records(words, text)
    got = [(i, length) for i, length, _, _ in m.match_loop(text)]
    assert got == list(zip(b["end_pos"].tolist(), b["length"].tolist()))


def test_fatal_error_convention():
    """Violated precondition: two-line message on stderr starting with the reference's banner
    (aho_corasick.c:24-36), then the calling thread exits."""
    code = ("import sys, os, threading; sys.path.insert(0, %r)\n"
            "import ctypes as C, aho_corasick_1975_amd as acm\n"
            "L = acm.lib(); m = acm.Machine(1)\n"
            "def violate():\n"
            "    cur = C.c_void_p(L.acm_initiate(m.handle))\n"
            "    L.acm_insert_end_of_keyword(C.byref(cur), None, None)\n"
            "    print('survived', flush=True)\n"
            "t = threading.Thread(target=violate, daemon=True); t.start(); t.join(5)\n"
            "print('main thread alive', flush=True); sys.stderr.flush(); os._exit(0)\n") % os.path.dirname(
                os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=120)
    err = p.stderr.decode()
    assert "FATAL ERROR: A prerequisite is not fulfilled in function acm_insert_end_of_keyword." in err
    assert "acm_insert_letter_of_keyword should be called first." in err
    assert b"survived" not in p.stdout          # the violating thread was terminated ...
    assert b"main thread alive" in p.stdout     # ... and only that thread (thrd_exit, not exit)


def test_get_keyword_rematerialises_match_holder():
    """acm_get_keyword(id) gives what acm_get_match gives for the same keyword: dictionary letters,
    length, value; ids are first-insertion ranks (duplicates keep the first rank and value)."""
    m = acm.Machine(1)
    vals = [C.c_size_t(100 + i) for i in range(6)]
    words = [b"bc", b"abc", b"bc", b"c", b"abc", b"zz"]
    for w, v in zip(words, vals):
        m.add_keyword(w, C.addressof(v))
    assert m.nb_keywords == 4
    distinct = [b"bc", b"abc", b"c", b"zz"]
    first_val = {b"bc": 100, b"abc": 101, b"c": 103, b"zz": 105}
    for kid, w in enumerate(distinct):
        word, value = m.keyword(kid)
        assert bytes(word) == w
        assert C.c_size_t.from_address(value).value == first_val[w]
    with pytest.raises(acm.ACMError):
        m.keyword(4)
    # the per-symbol loop reports the same (length, spelling, value) for every match
    for i, length, word, value in m.match_loop(b"xabczz"):
        kid = distinct.index(bytes(word))
        w2, v2 = m.keyword(kid)
        assert w2 == word and v2 == value and len(w2) == length
