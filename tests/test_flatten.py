"""The flattened tables (acm_flatten / acm_flat_dense_rows: what the GPU kernels consume) checked
on the CPU against the oracle with the scalar walkers of oracle/flat_walker.c."""
import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from oracle import pyoracle as po
from tests import flatwalk
from tests.brute import brute_records
from tests.cases import build_pair, build_pair_packed, small_cases

CASES = small_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_flat_tables_reproduce_oracle_records(name):
    kws, text, sym = CASES[name]
    m, o = build_pair(kws, sym)
    want = o.scan(text)
    flat = m.flatten()
    assert flat.info.n_keywords == o.nb_keywords and flat.info.n_states == o.nb_states
    assert flat.info.lmax == o.lmax
    got = flatwalk.walk_csr(flat, text)
    assert np.array_equal(got, want)
    if sym == 1:
        assert np.array_equal(flatwalk.walk_dense(flat, text), want)
        if flat.info.n_states <= 32768:
            assert np.array_equal(flatwalk.walk_dense(flat, text, entry_bytes=4), want)
    if len(text) <= 2000:
        assert np.array_equal(want, brute_records(kws, text))


def test_flat_structure_invariants():
    kws, text, _ = CASES["ternary_dense"]
    m, o = build_pair(kws, 1)
    f = m.flatten()
    n = f.info.n_states
    assert f.fail[0] == 0 and f.depth[0] == 0 and f.nb_outputs[0] == 0
    assert np.all(np.diff(f.depth.astype(np.int64)) >= 0)                     # BFS order: depth monotone
    assert np.all(f.fail[1:] < np.arange(1, n))                               # f(s) < s
    assert np.all(f.depth[f.fail[1:]] < f.depth[1:])
    term = (f.term_kw != 0xFFFFFFFF).astype(np.uint32)
    assert np.array_equal(f.nb_outputs[1:], term[1:] + f.nb_outputs[f.fail[1:]])   # reference :207
    assert sorted(f.term_kw[f.term_kw != 0xFFFFFFFF].tolist()) == list(range(f.info.n_keywords))
    assert np.array_equal(f.term_kw[f.kw_state], np.arange(f.info.n_keywords))
    for d in range(f.info.lmax + 1):
        sl = slice(f.depth_start[d], f.depth_start[d + 1])
        assert np.all(f.depth[sl] == d)
    assert f.depth_start[f.info.lmax + 1] == n
    assert f.info.max_outputs == f.nb_outputs.max()
    # rows sorted by symbol value
    for s in range(n):
        row = f.edge_sym[f.row_ptr[s]:f.row_ptr[s + 1]]
        assert np.all(np.diff(row.astype(np.int64)) > 0)


def test_keyword_ids_are_first_insertion_ranks():
    """keyword_id = nb_sequences just before aho_corasick.c:352; duplicates keep the first rank."""
    kws = [b"bc", b"abc", b"bc", b"c", b"abc", b"zz"]
    m, o = build_pair(kws, 1)
    f = m.flatten()
    got = flatwalk.walk_csr(f, b"abczz")
    want = [(2, 3, 1), (2, 2, 0), (2, 1, 2), (4, 2, 3)]
    assert [(int(r["end_pos"]), int(r["length"]), int(r["keyword_id"])) for r in got] == want
    assert np.array_equal(got, o.scan(b"abczz"))


def test_empty_machine_and_ineligible_symbol_size():
    m = acm.Machine(1)
    f = m.flatten()
    assert f.info.n_states == 1 and f.info.n_edges == 0 and f.info.lmax == 0
    assert flatwalk.walk_csr(f, b"anything").size == 0
    # symbol sizes other than 1, 2, 4, 8 bytes stay on the per-symbol API (raw C API: 3-byte symbols)
    import ctypes as C
    L = acm.lib()
    size3 = C.c_size_t(3)
    m3 = L.acm_create(C.c_void_p.in_dll(L, "ACM_CMP_DEFAULT"), C.cast(C.pointer(size3), C.c_void_p), None)
    letters = C.create_string_buffer(b"abcdef")
    cur = C.c_void_p(L.acm_initiate(m3))
    L.acm_insert_letter_of_keyword(C.byref(cur), C.cast(letters, C.c_void_p))
    L.acm_insert_end_of_keyword(C.byref(cur), None, None)
    flat = C.c_void_p()
    assert L.acm_flatten(m3, C.byref(flat)) == -1
    L.acm_release(m3)


def test_emit_from_and_pos_base_on_walkers():
    kws, text, _ = CASES["ternary_dense"]
    m, o = build_pair(kws, 1)
    f = m.flatten()
    full = o.scan(text)
    cut = 12345
    part = flatwalk.walk_dense(f, text, emit_from=cut, pos_base=1000)
    ref = full[full["end_pos"] >= cut].copy()
    ref["end_pos"] += 1000
    assert np.array_equal(part, ref)


@pytest.mark.parametrize("K,n", [(1000, 1 << 20)])
def test_synthetic_spec_matches_survey_table_sizes(K, n):
    """K=1,000 -> 6,492 states, per-depth histogram, Lmax 12."""
    kd, ko = acm.synth.keywords(K)
    m, o = build_pair_packed(kd, ko)
    f = m.flatten()
    assert f.info.n_states == 6492 and f.info.lmax == 12 and f.info.n_keywords == 1000
    assert np.bincount(f.depth).tolist() == [1, 26, 535, 979, 1000, 880, 769, 659, 537, 432, 336, 218, 120]
    text = acm.synth.text(n, kd, ko)
    want = o.scan(text)
    assert np.array_equal(flatwalk.walk_dense(f, text), want)
    assert np.array_equal(flatwalk.walk_csr(f, text), want)


def test_synthetic_64MiB_digest_matches_survey():
    """K=1,000, N=2^26 -> 35,453 matches, digest 75c631ca92f2fd08 (reference
    + survey harness).  Checked on the oracle (multi-threaded) and on the flat tables."""
    kd, ko = acm.synth.keywords(1000)
    n = 1 << 26
    text = acm.synth.text(n, kd, ko)
    m, o = build_pair_packed(kd, ko, variant=po.MEYER85)
    cnt, dig = o.scan_mt(text, 8)
    assert cnt == 35453 and dig == 0x75c631ca92f2fd08
    got = flatwalk.walk_dense(m.flatten(), text)
    assert got.size == 35453 and po.digest(got) == 0x75c631ca92f2fd08


def test_eight_byte_symbols_are_interned():
    """ACM_CMP_DEFAULT over 8-byte symbols: the flat tables carry 1 + rank of each symbol among the
    dictionary's distinct symbols (keys64); a text interned the same way (0 = any other symbol)
    walks to the oracle's records; spellings come back as the original 8-byte symbols."""
    rng = np.random.default_rng(3)
    vocab = rng.integers(0, 1 << 63, size=40, dtype=np.uint64) | np.uint64(1 << 40)
    kws = [vocab[rng.integers(0, 40, size=rng.integers(1, 6))] for _ in range(60)]
    m, o = build_pair(kws, 8)
    flat = m.flatten()
    assert flat.info.sym_bytes == 8 and flat.keys64 is not None
    assert np.all(np.diff(flat.keys64.astype(object)) > 0)
    used = np.unique(np.concatenate(kws))
    assert np.array_equal(flat.keys64, used)
    assert flat.edge_sym.min() >= 1 and flat.edge_sym.max() <= used.size
    text = np.concatenate([vocab[rng.integers(0, 40, size=3000)], rng.integers(0, 1 << 62, size=500, dtype=np.uint64)])
    rng.shuffle(text)
    ids = np.zeros(text.size, np.uint64)   # (the CPU walker reads symbols of the tables' width: 8 bytes)
    pos = np.searchsorted(flat.keys64, text)
    hit = (pos < flat.keys64.size) & (flat.keys64[np.minimum(pos, flat.keys64.size - 1)] == text)
    ids[hit] = pos[hit].astype(np.uint64) + 1
    want = o.scan(text)
    assert want.size > 20
    assert np.array_equal(flatwalk.walk_csr(flat, ids), want)
    seen = []
    for kw in kws:
        if kw.tolist() not in seen:
            seen.append(kw.tolist())
    for k, spelled in enumerate(seen):
        assert flat.keyword(k).tolist() == spelled
    # blob round trip with the symbol table
    from aho_corasick_1975_amd import binding
    back = binding.FlatTables.from_bytes(flat.to_bytes())
    assert np.array_equal(back.keys64, flat.keys64) and np.array_equal(back.edge_sym, flat.edge_sym)
    assert back.keyword(0).tolist() == seen[0]
    blob = bytearray(flat.to_bytes())
    blob[-8:] = blob[-16:-8]                # last two symbols equal: not ascending
    from tests.test_flat_blob import refnv
    with pytest.raises(RuntimeError):
        binding.FlatTables.from_bytes(refnv(bytes(blob)))
