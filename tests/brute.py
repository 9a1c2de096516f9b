"""Definition-level reference, independent of every automaton in this repo: a keyword matches
at end position i iff it is a suffix of text[:i+1]; matches at one position are listed longest
first (reference aho_corasick.c:459-466).  O(n * K): small inputs only."""
import numpy as np

from oracle.pyoracle import RECORD_DTYPE


def brute_records(keywords, text, eq=None):
    """keywords: list of sequences in first-insertion order (duplicates allowed; they collapse
    onto the first rank, reference aho_corasick.c:346-355).  eq: optional symbol equality."""
    text = list(text)
    distinct = []
    for kw in keywords:
        kw = list(kw)
        if not any(_same(kw, d, eq) for d in distinct):
            distinct.append(kw)
    recs = []
    for i in range(len(text)):
        here = []
        for rank, kw in enumerate(distinct):
            L = len(kw)
            if L <= i + 1 and _same(text[i + 1 - L:i + 1], kw, eq):
                here.append((i, L, rank))
        here.sort(key=lambda r: -r[1])
        recs.extend(here)
    return np.array(recs, dtype=RECORD_DTYPE) if recs else np.zeros(0, dtype=RECORD_DTYPE)


def _same(a, b, eq):
    if len(a) != len(b):
        return False
    if eq is None:
        return list(a) == list(b)
    return all(eq(x, y) for x, y in zip(a, b))
