"""Serialised flat tables (acm_flat_to_blob / from_blob / save / load / acm_flat_keyword):
round trips, rejection of damaged blobs, and keyword spellings read back from the tables."""
import struct

import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from aho_corasick_1975_amd import binding
from tests import flatwalk
from tests.cases import build_pair, small_cases

CASES = small_cases()
ARRAYS = ("row_ptr", "edge_sym", "edge_next", "fail", "depth", "nb_outputs", "term_kw", "out_link", "depth_start", "kw_state")
INFO = ("sym_bytes", "n_states", "n_keywords", "n_edges", "lmax", "max_outputs", "alpha_lo", "alpha_span", "width")
HEADER = 80
E_FORMAT = -8


def same_tables(a, b):
    return all(getattr(a.info, k) == getattr(b.info, k) for k in INFO) and all(
        np.array_equal(getattr(a, k), getattr(b, k)) for k in ARRAYS)


def refnv(blob):
    """Re-stamps the checksum so that damage to the payload reaches the structural checks."""
    h = 0xcbf29ce484222325
    for byte in blob[HEADER:]:
        h = ((h ^ byte) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return blob[:64] + struct.pack("<Q", h) + blob[72:]


@pytest.mark.parametrize("name", sorted(CASES))
def test_blob_round_trip(name, tmp_path):
    kws, text, sym = CASES[name]
    m, o = build_pair(kws, sym)
    flat = m.flatten()
    blob = flat.to_bytes()
    assert len(blob) == HEADER + 4 * sum(getattr(flat, k).size for k in ARRAYS)
    back = binding.FlatTables.from_bytes(blob)
    assert same_tables(flat, back)
    path = tmp_path / "dict.ac75"
    flat.save(path)
    assert path.read_bytes() == blob
    loaded = binding.FlatTables.load(path)
    assert same_tables(flat, loaded)
    # the loaded tables scan like the machine they came from
    assert np.array_equal(flatwalk.walk_csr(loaded, text), o.scan(text))
    # spellings from the tables alone == what was inserted (first insertion rank order)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[sym]
    seen = []
    for kw in kws:
        spelled = (np.frombuffer(kw, dtype=dt) if isinstance(kw, (bytes, bytearray)) else np.asarray(kw, dtype=dt)).tolist()
        if spelled and spelled not in seen:
            seen.append(spelled)
    assert len(seen) == loaded.info.n_keywords
    for k, spelled in enumerate(seen):
        assert loaded.keyword(k).tolist() == spelled


def test_empty_machine_blob():
    m = acm.Machine(1)
    flat = m.flatten()
    back = binding.FlatTables.from_bytes(flat.to_bytes())
    assert same_tables(flat, back) and back.info.n_states == 1 and back.info.n_keywords == 0


def test_synthetic_1k_blob_round_trip():
    kd, ko = acm.synth.keywords(1000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    flat = m.flatten()
    back = binding.FlatTables.from_bytes(flat.to_bytes())
    assert same_tables(flat, back) and back.info.n_states == 6492
    for k in (0, 1, 499, 999):
        assert back.keyword(k).tobytes() == kd[ko[k]:ko[k + 1]].tobytes()


def test_damaged_blobs_are_rejected():
    kws, _, _ = CASES["ternary_dense"]
    m, _ = build_pair(kws, 1)
    flat = m.flatten()
    blob = flat.to_bytes()
    n, E = flat.info.n_states, flat.info.n_edges

    def rejected(b):
        with pytest.raises(RuntimeError) as ei:
            binding.FlatTables.from_bytes(b)
        return "blob" in str(ei.value)

    assert rejected(b"")
    assert rejected(blob[:40])                                   # truncated header
    assert rejected(blob[:-4])                                   # truncated payload
    assert rejected(b"XX" + blob[2:])                            # magic
    assert rejected(blob[:8] + struct.pack("<I", 2) + blob[12:])  # version
    assert rejected(blob[:HEADER + 5] + bytes([blob[HEADER + 5] ^ 1]) + blob[HEADER + 6:])  # checksum
    # structurally wrong payloads with a valid checksum
    words = np.frombuffer(blob[HEADER:], dtype="<u4").copy()
    off = {}
    cur = 0
    for k in ARRAYS:
        off[k] = cur
        cur += getattr(flat, k).size

    def with_word(name, idx, value):
        w = words.copy()
        w[off[name] + idx] = value
        return refnv(blob[:HEADER] + w.tobytes())

    deep = n - 1
    assert rejected(with_word("fail", deep, (int(flat.fail[deep]) + 1) % n))      # wrong failure link
    assert rejected(with_word("fail", 1, n + 7))                                  # out of range
    assert rejected(with_word("edge_next", 0, 2))                                 # not breadth-first
    assert rejected(with_word("edge_sym", E - 1, 0x1FF))                          # not a byte
    assert rejected(with_word("row_ptr", 1, E + 3))                               # row beyond the edges
    assert rejected(with_word("nb_outputs", deep, int(flat.nb_outputs[deep]) + 1))
    assert rejected(with_word("depth", deep, int(flat.depth[deep]) + 1))
    assert rejected(with_word("out_link", deep, deep))
    assert rejected(with_word("kw_state", 0, 0))
    assert rejected(with_word("term_kw", int(flat.kw_state[0]), flat.info.n_keywords + 3))
    # a header that lies about the sizes
    hdr = bytearray(blob[:HEADER])
    hdr[20:24] = struct.pack("<I", n + 1)
    assert rejected(bytes(hdr) + blob[HEADER:])


def test_load_missing_file(tmp_path):
    with pytest.raises(RuntimeError):
        binding.FlatTables.load(tmp_path / "absent.ac75")
