"""Shared dictionaries/texts for the parity tests (CPU table tests and GPU tests use the same
cases so that a GPU failure can be bisected to the flattener or to the kernel)."""
import numpy as np

import aho_corasick_1975_amd as acm
from oracle import pyoracle as po


def build_pair(keywords, sym_size=1, variant=po.AC75):
    """Same keywords, same order, into the product machine and into the oracle."""
    m = acm.Machine(sym_size)
    o = po.Oracle(sym_size, variant)
    for kw in keywords:
        m.add_keyword(kw)
        o.add_keyword(kw)
    return m, o


def build_pair_packed(data, off, sym_size=1, variant=po.AC75):
    m = acm.Machine(sym_size)
    o = po.Oracle(sym_size, variant)
    m.add_keywords_packed(data, off)
    o.add_keywords_packed(data, off)
    return m, o


def rand_words(rng, n, lo, hi, minlen, maxlen, dtype=np.uint8):
    return [rng.integers(lo, hi, size=int(rng.integers(minlen, maxlen + 1))).astype(dtype) for _ in range(n)]


def small_cases():
    """name -> (keywords, text, sym_size): edge cases the reference's tests and SURVEY.md
    Appendix B call out (nested suffix keywords, duplicates, single symbol, full byte range,
    long keywords, symbols absent from the dictionary, dense overlaps)."""
    rng = np.random.default_rng(1234)
    c = {}
    c["readme"] = ([b"he", b"she", b"his", b"hers"], b"To ushers: he found his pencil, but she could not find hers.", 1)
    c["nested_suffixes"] = ([b"a", b"aa", b"aaa", b"aaaa", b"ba", b"baa"], b"aaaabaaaabaab" * 40, 1)
    c["duplicates"] = ([b"abc", b"bc", b"abc", b"c", b"bc", b"abcd"], b"xabcdabcabcdcbcabc" * 30, 1)
    c["single_symbol"] = ([b"z"], b"zzzazzz" * 100, 1)
    c["no_match"] = ([b"needle", b"pin"], bytes(rng.integers(65, 70, size=5000, dtype=np.uint8)), 1)
    c["binary_full_range"] = ([bytes(w) for w in rand_words(rng, 300, 0, 256, 1, 5)],
                              bytes(rng.integers(0, 256, size=40000, dtype=np.uint8)), 1)
    c["ternary_dense"] = ([bytes(w) for w in rand_words(rng, 400, 97, 100, 1, 9)],
                          bytes(rng.integers(97, 100, size=30000, dtype=np.uint8)), 1)
    c["long_keywords"] = ([bytes(w) for w in rand_words(rng, 40, 97, 99, 20, 70)] + [b"ab", b"ba"],
                          bytes(rng.integers(97, 99, size=50000, dtype=np.uint8)), 1)
    c["text_outside_alphabet"] = ([b"mno", b"nop", b"o"], bytes(rng.integers(0, 256, size=20000, dtype=np.uint8)), 1)
    c["u16_symbols"] = (rand_words(rng, 200, 0, 600, 1, 5, np.uint16), rng.integers(0, 600, size=30000).astype(np.uint16), 2)
    c["u32_symbols"] = (rand_words(rng, 300, 0, 70000, 1, 4, np.uint32), rng.integers(0, 70000, size=30000).astype(np.uint32), 4)
    c["u32_byteorder"] = ([np.array([0x01000000, 0x00000001], np.uint32), np.array([0x00000100], np.uint32),
                           np.array([0x00010000, 0x00000100], np.uint32)],
                          rng.choice(np.array([1, 0x100, 0x10000, 0x1000000], np.uint32), size=5000), 4)
    return c
