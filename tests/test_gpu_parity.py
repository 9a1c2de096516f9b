"""GPU parity: the HIP bulk scan (through the C ABI of include/acm_gpu.h) against the CPU oracle's
caller loop on the same inputs -- bit-exact records in canonical order."""
import ctypes as C
import os

import numpy as np
import pytest

import aho_corasick_1975_amd as acm
from oracle import pyoracle as po
from tests.cases import build_pair, build_pair_packed, small_cases

pytestmark = pytest.mark.gpu

CASES = small_cases()


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the GPU box)"
    torch.cuda.set_device(0)
    return torch


def _dev(torch, arr):
    a = np.frombuffer(bytes(arr), dtype=np.uint8) if isinstance(arr, (bytes, bytearray)) else np.ascontiguousarray(arr)
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    elif a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a.copy()).cuda()


def test_native_library_is_loaded_and_sees_the_gpu(torch_cuda):
    assert acm.lib().acm_gpu_device_count() >= 1


@pytest.mark.parametrize("name", sorted(CASES))
def test_small_cases_bit_exact(torch_cuda, name):
    kws, text, sym = CASES[name]
    m, o = build_pair(kws, sym)
    want = o.scan(text)
    plan = m.plan(0)
    got = plan.scan_sorted(_dev(torch_cuda, text))
    assert got.size == want.size
    assert np.array_equal(got, want)
    # count-only entry point == sum of acm_match (generic_test.c:272-273)
    cnt = plan.count(_dev(torch_cuda, text))
    assert int(cnt.item()) == o.count(text) == want.size
    # host-buffer C ABI path (no torch involved)
    assert np.array_equal(plan.scan_host(np.frombuffer(bytes(text), np.uint8) if isinstance(text, bytes) else text), want)


@pytest.mark.parametrize("name", sorted(k for k in CASES if CASES[k][2] > 1))
def test_wide_symbol_cases_sparse_walk(torch_cuda, monkeypatch, name):
    """2- and 4-byte symbols through the sparse automaton walk (the default is the start-parallel
    kernel, covered by test_small_cases_bit_exact)."""
    monkeypatch.setenv("ACM_GPU_SPARSE", "walk")
    kws, text, sym = CASES[name]
    m, o = build_pair(kws, sym)
    plan = m.plan(0)
    assert plan.info.kernel == 3
    want = o.scan(text)
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text)), want)
    assert int(plan.count(_dev(torch_cuda, text)).item()) == want.size


def test_acm_scan_on_machine_follows_dictionary_updates(torch_cuda):
    """acm_scan() caches a plan inside the machine and rebuilds it when keywords were added."""
    m, o = build_pair([b"he", b"she"], 1)
    text = b"ushers and heroes; she sells hers" * 50
    assert np.array_equal(m.scan_host(text), o.scan(text))
    for w in (b"his", b"hers", b"s"):
        m.add_keyword(w)
        o.add_keyword(w)
    assert np.array_equal(m.scan_host(text), o.scan(text))


@pytest.mark.parametrize("K", [300, 20000])
def test_plan_with_delta_scans_counts_and_streams(torch_cuda, K):
    """acm_gpu_plan_update on a dense plan (300 keywords) and on a 4-gram plan (20,000): the keywords
    added afterwards live in a delta plan; whole scans, count-only scans, shards with warm-up and a
    stream fed in pieces all report the union, with the machine's keyword ids."""
    kd, ko = acm.synth.keywords(K + 150)
    m, o = build_pair_packed(kd[:ko[K]], ko[:K + 1], variant=po.MEYER85)
    plan = m.plan(0)
    base_kernel = plan.info.kernel
    text = acm.synth.text(1 << 20, kd, ko)
    dev = _dev(torch_cuda, text)
    assert np.array_equal(plan.scan_sorted(dev), o.scan(text))
    for k in range(K, K + 150):
        w = kd[ko[k]:ko[k + 1]]
        m.add_keyword(w)
        o.add_keyword(w)
        if k % 50 == 49 or k == K:
            plan.update(m)
            assert plan.info.kernel == base_kernel and plan.info.merges == 0 and plan.info.delta_keywords == k + 1 - K
            want = o.scan(text)
            assert np.array_equal(plan.scan_sorted(dev), want)
            assert int(plan.count(dev).item()) == want.size
    want = o.scan(text)
    assert int(np.sum(want["keyword_id"] >= K)) > 10          # the new keywords do match
    for b, e in ((5, 77), (4103, 4103 + 50000), (text.size - 33, text.size)):
        rb = max(b - (m.lmax - 1), 0)
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=rb)
        assert np.array_equal(got, want[(want["end_pos"] >= b) & (want["end_pos"] < e)])
    st = plan.stream(max_piece_symbols=100000, record_capacity=want.size + 16)
    for off in range(0, text.size, 77777):
        st.feed(text[off:off + 77777])
    assert np.array_equal(st.finish(), want)
    st.close()
    assert np.array_equal(plan.scan_host(text[:50000]), want[want["end_pos"] < 50000])
    # enough new keywords and the next update merges everything into one plan again
    kd2, ko2 = acm.synth.keywords(K + 150 + max(256, K // 8) + 10)
    for k in range(K + 150, len(ko2) - 1):
        w = kd2[ko2[k]:ko2[k + 1]]
        m.add_keyword(w)
        o.add_keyword(w)
    plan.update(m)
    assert plan.info.merges == 1 and plan.info.delta_keywords == 0
    assert np.array_equal(plan.scan_sorted(dev), o.scan(text))


def test_delta_given_up_after_enough_second_passes(torch_cuda):
    """A delta plan makes every scan two passes over the text (0.57 against 0.32 ms per GiB on config
    2's dictionary): once the texts scanned with it add up to more than one plan of everything
    costs (14 Gi symbols per 1,000 keywords) the next acm_gpu_plan_update merges, new keywords or
    not.  Config 2's dictionary, one more keyword, 1 GiB scanned 15 times."""
    kd, ko = acm.synth.keywords(1001)
    m = acm.Machine(1)
    m.add_keywords_packed(kd[:ko[1000]], ko[:1001])
    plan = m.plan(0)
    n = 1 << 30
    text = acm.synth.device_text(n, kd[:ko[1000]], ko[:1001])
    m.add_keyword(kd[ko[1000]:ko[1001]])
    plan.update(m)
    assert plan.info.delta_keywords == 1 and plan.info.merges == 0
    rec = torch_cuda.empty((700000, 2), dtype=torch_cuda.int64, device="cuda")
    cnt = torch_cuda.zeros(1, dtype=torch_cuda.int64, device="cuda")
    for _ in range(13):
        plan.scan(text, records=rec, count=cnt)
    with_delta = acm.synth.device_digest(rec, int(cnt.item()))
    plan.update(m)
    assert plan.info.delta_keywords == 1 and plan.info.merges == 0      # 13 Gi symbols: not yet
    for _ in range(2):
        plan.scan(text, records=rec, count=cnt)
    plan.update(m)
    assert plan.info.delta_keywords == 0 and plan.info.merges == 1      # 15 Gi: one plan of all 1,001 keywords
    plan.scan(text, records=rec, count=cnt)
    assert acm.synth.device_digest(rec, int(cnt.item())) == with_delta
    assert with_delta[0] >= 555000


def test_delta_plan_that_is_a_4gram_plan_itself(torch_cuda, monkeypatch):
    """2,400 keywords added to a plan of 20,000: the delta is big enough for the 4-gram kernel, whose
    tables hold keyword ids (hits that carry their keyword) -- they must be the machine's ids, i.e.
    offset by the base plan's 20,000."""
    monkeypatch.setenv("ACM_GPU_GRAM", "2")                    # (the 4-gram kernel whatever the size: the delta's too)
    K, D = 20000, 2400
    kd, ko = acm.synth.keywords(K + D)
    m, o = build_pair_packed(kd[:ko[K]], ko[:K + 1], variant=po.MEYER85)
    plan = m.plan(0)
    assert plan.info.kernel == 5
    for k in range(K, K + D):
        w = kd[ko[k]:ko[k + 1]]
        m.add_keyword(w)
        o.add_keyword(w)
    plan.update(m)
    assert plan.info.merges == 0 and plan.info.delta_keywords == D
    text = acm.synth.text(1 << 20, kd, ko)
    want = o.scan(text)
    new4 = want[(want["keyword_id"] >= K) & (want["length"] == 4)]
    assert new4.size > 100                                     # 4-symbol keywords of the delta do match
    dev = _dev(torch_cuda, text)
    assert np.array_equal(plan.scan_sorted(dev), want)
    assert int(plan.count(dev).item()) == want.size


def _novel_words(novel_bytes):
    """generic_test.c:191-197: letters in lower case, everything else a blank"""
    t = np.frombuffer(novel_bytes, np.uint8).copy()
    upper = (t >= 65) & (t <= 90)
    t[upper] += 32
    t[~((t >= 97) & (t <= 122))] = 32
    return t


def test_dictionary_built_up_while_scanning_the_novel(torch_cuda, novel_bytes):
    """The reference's second test (generic_test.c:166-239) on bytes: the dictionary is built up
    from the text it is scanning -- every word met for the first time goes in as " word " -- and
    matching goes on between the inserts, here through acm_scan() on the machine (the cached
    plan follows the machine: acm_gpu_plan_update takes the new keywords as a delta plan and
    rebuilds only now and then).  After EVERY insert a window of the text around the new word is
    scanned and compared with the oracle holding the same dictionary; in the end 6,966 keywords
     and the whole text against the oracle."""
    text = _novel_words(novel_bytes)
    m, o = build_pair([], 1, variant=po.MEYER85)
    import re
    seen, inserts = set(), 0
    raw = text.tobytes()
    for mt in re.finditer(rb"[a-z]+", raw):
        w = mt.group(0)
        if w in seen:
            continue
        seen.add(w)
        kw = b" " + w + b" "
        m.add_keyword(kw)
        o.add_keyword(kw)
        inserts += 1
        lo, hi = max(mt.start() - 1500, 0), min(mt.end() + 1500, len(raw))
        window = text[lo:hi]
        got = m.scan_host(window)
        want = o.scan(window)
        assert got.size == want.size and np.array_equal(got, want), (inserts, w)
    assert inserts == m.nb_keywords == 6966
    want = o.scan(text)
    assert np.array_equal(m.scan_host(text), want)


def test_config1_novel(torch_cuda, novel_bytes):
    """BASELINE config 1 on the GPU path: 11,676 records, identical to the oracle's."""
    m, o = build_pair([b"he", b"she", b"his", b"hers"], 1)
    got = m.plan(0).scan_sorted(_dev(torch_cuda, novel_bytes))
    assert got.size == 11676
    assert np.array_equal(got, o.scan(novel_bytes))
    assert np.bincount(got["keyword_id"]).tolist() == [9513, 1273, 784, 106]


@pytest.mark.parametrize("n", [0, 1, 5, 15, 16, 17, 8191, 8192, 8193, 16384 + 7, 3 * 8192 + 100, 1 << 20])
def test_ragged_lengths_and_tile_seams(torch_cuda, n):
    """Head/tail ranges and tile seams of the dense kernel: every length around the 8 KiB tile."""
    kd, ko = acm.synth.keywords(300)
    m, o = build_pair_packed(kd, ko)
    text = acm.synth.text(((n + 4095) // 4096 + 1) * 4096, kd, ko)[:n]
    plan = m.plan(0)
    got = plan.scan_sorted(_dev(torch_cuda, text)) if n else plan.scan_host(text)
    assert np.array_equal(got, o.scan(text))


def test_keywords_across_every_seam(torch_cuda):
    """Plant one keyword across each chunk boundary (64 B) of a few tiles."""
    kws = [b"abcdefghijkl", b"ghij", b"l", b"kla"]
    m, o = build_pair(kws, 1)
    n = 4 * 8192 + 1000
    text = np.full(n, ord("x"), dtype=np.uint8)
    for p in range(58, n - 12, 64):
        text[p:p + 12] = np.frombuffer(b"abcdefghijkl", np.uint8)
    got = m.plan(0).scan_sorted(_dev(torch_cuda, text))
    assert np.array_equal(got, o.scan(text))


def test_emit_from_and_pos_base(torch_cuda):
    kd, ko = acm.synth.keywords(300)
    m, o = build_pair_packed(kd, ko)
    text = acm.synth.text(1 << 18, kd, ko)
    full = o.scan(text)
    plan = m.plan(0)
    for cut in (0, 11, 4096, 100001):
        got = plan.scan_sorted(_dev(torch_cuda, text), emit_from=cut, pos_base=1 << 40)
        ref = full[full["end_pos"] >= cut].copy()
        ref["end_pos"] += np.uint64(1 << 40)
        assert np.array_equal(got, ref)


def test_overflow_is_reported_not_silent(torch_cuda):
    m, o = build_pair([b"a", b"aa"], 1)
    text = np.full(100000, ord("a"), dtype=np.uint8)
    plan = m.plan(0)
    rec, cnt = plan.scan(_dev(torch_cuda, text), capacity=1000)
    torch_cuda.cuda.synchronize()
    assert int(cnt.item()) == 199999 > rec.shape[0]
    out = np.zeros(10, dtype=acm.RECORD_DTYPE)
    n = C.c_uint64(0)
    rc = acm.lib().acm_gpu_scan_host(plan.h, text.ctypes.data, text.size, 0, 0, out.ctypes.data, 10, C.byref(n))
    assert rc == -4 and n.value == 199999
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text)), o.scan(text))


@pytest.mark.parametrize("shape", ["dense", "sparse", "crowded_in_sparse", "one_position", "tiny"])
def test_order_records_by_buckets_equals_a_plain_sort(torch_cuda, shape):
    """acm_gpu_order_records_device (position buckets, wave sorts, counting sorts) on record sets it
    has to get right whatever a scan leaves: dense (every bucket a counting or wave sort), sparse
    (windows of buckets), a crowded stretch inside a sparse set, many records at ONE position
    (longest first), a handful.  Against numpy's sort of the same records; the radix fallback
    (ACM_GPU_ORDER=radix is read per call) must give the same."""
    torch = torch_cuda
    rng = np.random.default_rng(sum(map(ord, shape)))
    m, _ = build_pair([b"a" * k for k in range(1, 13)], 1)      # lmax 12: lengths 1..12 are valid
    plan = m.plan(0)
    pos_lo, span = 1 << 33, 1 << 24
    if shape == "dense":
        pos = rng.integers(0, span, size=300000)
    elif shape == "sparse":
        pos = rng.integers(0, span, size=3000)
    elif shape == "crowded_in_sparse":
        pos = np.concatenate([rng.integers(0, span, size=2000), rng.integers(5 << 12, (5 << 12) + 3000, size=1500),
                              rng.integers(span - 700, span, size=900)])
    elif shape == "one_position":
        pos = np.concatenate([np.full(12, 777), rng.integers(0, span, size=40), np.full(12, span - 1)])
    else:
        pos = np.array([5, 3, 5, 9, 0])
    # distinct (position, length) pairs: a keyword is its end and its length
    pairs = np.unique(np.stack([pos, rng.integers(1, 13, size=pos.size)], axis=1), axis=0)
    if shape == "one_position":
        pairs = np.unique(np.concatenate([pairs, np.stack([np.full(12, 777), np.arange(1, 13)], axis=1),
                                          np.stack([np.full(12, span - 1), np.arange(1, 13)], axis=1)]), axis=0)
    rng.shuffle(pairs)
    n = pairs.shape[0]
    rec = np.zeros(n, dtype=acm.RECORD_DTYPE)
    rec["end_pos"] = pairs[:, 0] + pos_lo
    rec["length"] = pairs[:, 1]
    rec["keyword_id"] = rng.integers(0, 1 << 30, size=n)
    want = rec[np.lexsort((-rec["length"].astype(np.int64), rec["end_pos"]))]
    for mode in ("buckets", "radix"):
        if mode == "radix":
            os.environ["ACM_GPU_ORDER"] = "radix"
        try:
            dev = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.int64).reshape(n, 2).copy()).cuda()
            plan.sort(dev, n, pos_lo, span)
            got = np.frombuffer(dev.cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE)
        finally:
            os.environ.pop("ACM_GPU_ORDER", None)
        assert np.array_equal(got, want), (shape, mode)
    plan.status()


@pytest.mark.parametrize("kind", ["dense", "gram", "starts16"])
def test_scan_ordered_one_call_equals_scan_then_order(torch_cuda, monkeypatch, kind):
    """acm_gpu_scan_ordered_device: scan and canonical order queued in one call (the order passes read
    the count on the device; both pass-C kernels are launched and the set's kind decides on the
    device which of them works).  Sparse and dense record sets, no records at all, one record, a
    buffer that is too small (total reported, repeat with room), the radix switch, against the
    oracle and against the two-call path."""
    torch = torch_cuda
    rng = np.random.default_rng(77 + len(kind))
    if kind == "dense":
        kws = [bytes(rng.integers(97, 123, size=rng.integers(3, 9)).astype(np.uint8)) for _ in range(300)]
        sym = 1
        texts = [rng.integers(97, 123, size=1 << 20).astype(np.uint8),          # sparse: a record per few thousand symbols
                 rng.integers(97, 100, size=200000).astype(np.uint8)]
        kws += [b"ab", b"abc", b"ca", b"b"]                                     # ... and the second text is full of these
    elif kind == "gram":
        kws = [rng.integers(97, 104, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(12000)]
        sym = 1
        texts = [rng.integers(96, 105, size=700000).astype(np.uint8), rng.integers(110, 120, size=50000).astype(np.uint8)]
    else:
        kws = [rng.integers(0, 400, size=rng.integers(1, 7)).astype(np.uint16) for _ in range(1500)]
        sym = 2
        texts = [rng.integers(0, 403, size=300000).astype(np.uint16), rng.integers(0, 60000, size=100000).astype(np.uint16)]
    m, o = build_pair(kws, sym)
    plan = m.plan(0)
    for text in texts:
        want = o.scan(text)
        dev = _dev(torch, text)
        for base in (0, (1 << 41) + 12345):
            w = want.copy()
            w["end_pos"] += base
            got = plan.scan_sorted(dev, pos_base=base, capacity=max(want.size, 1))
            assert np.array_equal(got, w), (kind, base)
            assert np.array_equal(plan.scan_sorted(dev, pos_base=base, capacity=max(want.size, 1), fused=False), w)
        if want.size > 3:       # too small: the total comes back, nothing is claimed about the buffer; then with room
            rec, cnt, _ = plan.scan_ordered(dev, capacity=want.size // 2)
            assert int(cnt.item()) == want.size
            plan.status()
            assert np.array_equal(plan.scan_sorted(dev, capacity=want.size // 2), want)
        cut = text.size - 5     # (almost) nothing to report
        tail = want[want["end_pos"] >= cut]
        assert np.array_equal(plan.scan_sorted(dev, emit_from=cut, capacity=64), tail)
        monkeypatch.setenv("ACM_GPU_ORDER", "radix")
        assert np.array_equal(plan.scan_sorted(dev, capacity=max(want.size, 1) + 100), want)
        monkeypatch.delenv("ACM_GPU_ORDER")
    # scratch that is too small, a missing count: refused, nothing queued
    L = acm.binding.lib()
    dev = _dev(torch, texts[0])
    rec = torch.empty((1000, 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    need = L.acm_gpu_scan_ordered_tmp_bytes(plan.h, 1000, texts[0].size)
    assert need > 16000
    tmp = torch.empty(need, dtype=torch.uint8, device="cuda")
    args = (plan.h, dev.data_ptr(), texts[0].size, 0, 0, rec.data_ptr(), 1000, cnt.data_ptr(), tmp.data_ptr())
    assert L.acm_gpu_scan_ordered_device(*args, need - 1, None) == acm.binding.ACM_GPU_E_ARG
    assert L.acm_gpu_scan_ordered_device(plan.h, dev.data_ptr(), texts[0].size, 0, 0, rec.data_ptr(), 1000, None, tmp.data_ptr(), need, None) == acm.binding.ACM_GPU_E_ARG
    assert L.acm_gpu_scan_ordered_device(*args, need, None) == 0
    torch.cuda.synchronize()
    plan.status()
    # a text without any match, and one with exactly one
    blank = np.full(5000, 255 if sym == 1 else 65535, dtype=texts[0].dtype)
    assert plan.scan_sorted(_dev(torch, blank)).size == 0
    one = blank.copy()
    one[1234:1234 + len(kws[0])] = np.frombuffer(kws[0], dtype=np.uint8) if isinstance(kws[0], bytes) else kws[0]
    assert np.array_equal(plan.scan_sorted(_dev(torch, one)), o.scan(one))


def test_tiled_order_every_position_full_of_matches(torch_cuda, monkeypatch):
    """The tiled scan's ordering pass (dev_tiles.h) on the densest texts a 4-gram plan can see: nested
    keywords a^4 .. a^12 in a text of a's -- nine records per position, a bucket of 2,048 positions
    holds 18,000: the path that streams the tile; the same with stretches of other letters between
    (buckets of a few hundred to two thousand records: sorted in registers where they lie); both
    against the oracle, whole and from a cut, and against the three general passes."""
    monkeypatch.setenv("ACM_GPU_GRAM", "2")
    kws = [b"a" * k for k in range(4, 13)] + [b"abab", b"baba", b"aabb"]
    m, o = build_pair(kws, 1)
    plan = m.plan(0)
    assert plan.info.kernel == 5 and plan.info.records_direct == 1
    rng = np.random.default_rng(99)
    solid = np.full(60000, ord("a"), dtype=np.uint8)
    mixed = solid.copy()
    for at in rng.integers(0, mixed.size - 40, size=900):       # islands of b's: the runs of a's get shorter
        mixed[at:at + int(rng.integers(1, 40))] = ord("b")
    for text in (solid, mixed):
        want = o.scan(text)
        dev = _dev(torch_cuda, text)
        got = plan.scan_sorted(dev, capacity=want.size)
        assert got.size == want.size and np.array_equal(got, want)
        cut = 33333
        assert np.array_equal(plan.scan_sorted(dev, emit_from=cut, pos_base=1 << 40, capacity=want.size),
                              _shifted(want[want["end_pos"] >= cut], 1 << 40))
        monkeypatch.setenv("ACM_GPU_ORDER", "buckets")
        assert np.array_equal(plan.scan_sorted(dev, capacity=want.size), want)
        monkeypatch.delenv("ACM_GPU_ORDER")


def _shifted(rec, by):
    r = rec.copy()
    r["end_pos"] += by
    return r


def test_tiled_scratch_is_an_upper_bound_for_every_emit_from(torch_cuda, monkeypatch):
    """acm_gpu_scan_ordered_tmp_bytes must cover the tiled layout of EVERY emit_from: the tile length
    is picked from the groups behind emit_from, so a cut a little into the text can give MORE tiles
    than emit_from = 0 (128 MiB on 256 CUs: R = 8 -> 16,384 tiles whole, R = 7 -> 18,725 from a cut of
    ~1 KiB).  Exactly that many bytes of scratch with a canary behind them; the records from the cut
    = the whole scan's from there on, and the canary is untouched."""
    torch = torch_cuda
    monkeypatch.setenv("ACM_GPU_GRAM", "2")
    rng = np.random.default_rng(2024)
    kws = [rng.integers(97, 123, size=rng.integers(4, 10)).astype(np.uint8) for _ in range(5000)]
    m, _ = build_pair(kws, 1)
    plan = m.plan(0)
    assert plan.info.kernel == 5 and plan.info.records_direct == 1
    n = 128 << 20
    dev = torch.randint(97, 123, (n,), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    whole_n = int(plan.count(dev).item())
    cap = whole_n + 16
    L = acm.binding.lib()
    need = L.acm_gpu_scan_ordered_tmp_bytes(plan.h, cap, n)
    canary = 1 << 20
    tmp = torch.full((need + canary,), 0xA5, dtype=torch.uint8, device="cuda")
    rec = torch.empty((cap, 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rec0, cnt0, _ = plan.scan_ordered(dev, capacity=cap)
    assert int(cnt0.item()) == whole_n
    for cut in (1100, 5 * 1024 + 3, (n // 5) + 77, n - 4096):
        rc = L.acm_gpu_scan_ordered_device(plan.h, dev.data_ptr(), n, cut, 0, rec.data_ptr(), cap, cnt.data_ptr(), tmp.data_ptr(), need, None)
        assert rc == 0
        k = int(cnt.item())
        plan.status()
        assert bool((tmp[need:] == 0xA5).all().item()), "scratch written past the size the query gave (cut %d)" % cut
        first = int((rec0[:whole_n, 0] < cut).sum().item())
        assert k == whole_n - first
        assert bool(torch.equal(rec[:k], rec0[first:whole_n])), "cut %d" % cut


def test_tiled_scan_far_more_records_than_the_raw_area_holds(torch_cuda, monkeypatch):
    """A tiled scan into a buffer that is smaller than the record set by more than the raw area's
    slack (one chunk per wave + 1: ~4.2 M slots on 256 CUs): the chunks beyond it are dropped with
    their links, so the size pass must not follow them -- it reports the exact total (summed in 64
    bits) and nothing else happens.  Then with room: the oracle's records."""
    torch = torch_cuda
    monkeypatch.setenv("ACM_GPU_GRAM", "2")
    rng = np.random.default_rng(515)
    kws = [rng.integers(97, 104, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(12000)]
    text = rng.integers(96, 105, size=(40 << 20) + 321).astype(np.uint8)
    m, o = build_pair(kws, 1)
    plan = m.plan(0)
    assert plan.info.kernel == 5 and plan.info.records_direct == 1
    want_n, want_d = o.scan_mt(text, 8)
    slack = (plan.info.grid_blocks * 16 + 1) * 1024
    assert want_n > 4096 + slack + 100000, (want_n, slack)
    dev = _dev(torch, text)
    for cap in (4096, 1 << 20):
        rec, cnt, _ = plan.scan_ordered(dev, capacity=cap)
        assert int(cnt.item()) == want_n, cap
        plan.status()
    rec, cnt, _ = plan.scan_ordered(dev, capacity=want_n)
    assert acm.synth.device_digest(rec, int(cnt.item())) == (want_n, want_d)
    ln = rec[:want_n, 1] & 0xFFFFFFFF
    ok = (rec[1:want_n, 0] > rec[:want_n - 1, 0]) | ((rec[1:want_n, 0] == rec[:want_n - 1, 0]) & (ln[1:] < ln[:-1]))
    assert bool(ok.all().item())
    plan.status()


def test_dense_matches_everywhere(torch_cuda):
    """Output blow-up: nested keywords matching at every position (queue flush path)."""
    kws = [b"a" * k for k in range(1, 9)] + [b"ab", b"b"]
    m, o = build_pair(kws, 1)
    rng = np.random.default_rng(5)
    text = rng.choice(np.frombuffer(b"aaab", np.uint8), size=200000)
    assert np.array_equal(m.plan(0).scan_sorted(_dev(torch_cuda, text)), o.scan(text))


def test_dense_region_overflow_many_times_per_wave(torch_cuda, monkeypatch):
    """The dense kernel's rare path, deterministically: a wave whose item region cannot take the
    next 16-step block expands it in place (region_make_room -> flush_queue reads back what other
    lanes of the wave have just parked) and starts the region over.  On a one-block grid
    (ACM_GPU_GRID_BLOCKS=1: 16 waves) a 16 MiB text with a match at every second symbol makes
    every wave fill its 4,352-item region about 250 times -- the lost-records race a work-in-
    progress build of round 2 showed on one small case in 70 of 300 runs would lose records on
    every run here.  Count and digest against the oracle; count-only pass too."""
    monkeypatch.setenv("ACM_GPU_GRID_BLOCKS", "1")
    m, o = build_pair([b"ab", b"b", b"bab"], 1)
    n = 16 << 20
    text = np.tile(np.frombuffer(b"ab", np.uint8), n // 2)
    plan = m.plan(0)
    assert plan.info.kernel == 1 and plan.info.grid_blocks == 1
    want_n, want_d = o.scan_mt(text, 8)
    assert want_n == 3 * (n // 2) - 1            # every b ends "b" and "ab", all but the first "bab"
    dev = _dev(torch_cuda, text)
    rec, cnt = plan.scan(dev, capacity=want_n + 16)
    assert acm.synth.device_digest(rec, int(cnt.item())) == (want_n, want_d)
    assert int(plan.count(dev).item()) == want_n
    plan.status()
    # and the same with the regular grid
    monkeypatch.delenv("ACM_GPU_GRID_BLOCKS")
    m2, _ = build_pair([b"ab", b"b", b"bab"], 1)
    rec, cnt = m2.plan(0).scan(dev, capacity=want_n + 16)
    assert acm.synth.device_digest(rec, int(cnt.item())) == (want_n, want_d)


@pytest.mark.parametrize("kind", ["dense", "gram2", "gram", "starts"])
def test_small_grids_every_part_of_the_tile_pool_has_a_block(torch_cuda, monkeypatch, kind):
    """Grids of fewer blocks than the tile pool has parts (a partitioned device, ACM_GPU_GRID_BLOCKS):
    the last sixteenth of a launch's tiles is a pool in up to 16 parts, and block b draws from part
    b * parts / blocks -- with 16 parts and fewer than 16 blocks some parts had no block at all and
    their tiles were never scanned (a work-in-progress build of round 3 lost exactly 15/256 of the
    records on a one-block grid; fixed by parts = min (16, blocks)).  Grids of 1, 3, 15, 17 and 20
    blocks, texts long enough for a pool, every kernel family that has one, against the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(616)
    if kind == "dense":
        kws = [bytes(rng.integers(97, 123, size=int(rng.integers(3, 9)), dtype=np.uint8)) for _ in range(300)] + [b"ab", b"q"]
        sym, text, env, kernel = 1, rng.integers(97, 123, size=(24 << 20) + 5, dtype=np.uint8), {}, 1
    elif kind in ("gram2", "gram"):
        kws = [rng.integers(97, 104, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(3000)]
        sym, text, kernel = 1, rng.integers(96, 105, size=(12 << 20) + 77).astype(np.uint8), 5
        env = {"ACM_GPU_GRAM": "2"} if kind == "gram2" else {"ACM_GPU_GRAM": "2", "ACM_GPU_GRAM2": "0"}
    else:
        kws = [rng.integers(0, 3000, size=rng.integers(1, 6)).astype(np.uint32) for _ in range(2000)]
        sym, text, env, kernel = 4, rng.integers(0, 3003, size=(6 << 20) + 3).astype(np.uint32), {}, 4
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _, o = build_pair(kws, sym)
    want_n, want_d = o.scan_mt(text, 8)
    assert want_n > 10000
    dev = _dev(torch, text)
    for blocks in (1, 3, 15, 17, 20):
        monkeypatch.setenv("ACM_GPU_GRID_BLOCKS", str(blocks))
        m, _ = build_pair(kws, sym)
        plan = m.plan(0)
        assert plan.info.kernel == kernel and plan.info.grid_blocks in (blocks, blocks * 16), (kind, plan.describe())
        if kind.startswith("gram"):
            assert plan.info.variant == (2 if kind == "gram2" else 0)
        rec, cnt = plan.scan(dev, capacity=want_n + 16)
        assert acm.synth.device_digest(rec, int(cnt.item())) == (want_n, want_d), (kind, blocks)
        assert int(plan.count(dev).item()) == want_n, (kind, blocks)
        plan.status()
        del rec, cnt, plan, m


@pytest.mark.parametrize("mode", ["sticky", "gram"])
def test_rows_colder_than_lds(torch_cuda, monkeypatch, mode):
    """A dictionary whose rows do not all fit in LDS: transitions through HBM-resident rows (sticky
    dense walk), or the 4-gram sieve kernel that such dictionaries get by default."""
    if mode == "sticky":
        monkeypatch.setenv("ACM_GPU_GRAM", "0")
    rng = np.random.default_rng(9)
    kws = [bytes(rng.integers(97, 123, size=int(rng.integers(4, 13)), dtype=np.uint8)) for _ in range(6000)]
    m, o = build_pair(kws, 1)
    plan = m.plan(0)
    assert plan.info.kernel == (1 if mode == "sticky" else 5) and plan.info.lds_rows < plan.info.dense_rows
    text = rng.integers(97, 123, size=1 << 20, dtype=np.uint8)
    for p in range(100, text.size - 16, 997):          # make deep states common
        w = kws[int(rng.integers(0, len(kws)))]
        text[p:p + len(w)] = np.frombuffer(w, np.uint8)
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text)), o.scan(text))


def test_config2_shape_64MiB_digest(torch_cuda):
    """BASELINE config 2's dictionary (1k keywords) on a 64 MiB prefix of its text: count and
    order-independent digest of SURVEY.md App. C (35,453 / 75c631ca92f2fd08), text generated on
    the device by acm_gpu_synth_text."""
    kd, ko = acm.synth.keywords(1000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    n = 1 << 26
    text = acm.synth.device_text(n, kd, ko)
    assert np.array_equal(text[:1 << 16].cpu().numpy(), acm.synth.text(1 << 16, kd, ko))
    got = m.plan(0).scan_sorted(text)
    assert got.size == 35453 and po.digest(got) == 0x75c631ca92f2fd08
    assert np.all(np.diff(got["end_pos"].astype(np.int64)) >= 0)


def test_config2_full_size_properties(torch_cuda):
    """Full 1 GiB of config 2: size-independent properties -- the count is additive over
    4096-aligned shards scanned with lmax-1 warm-up, records are sorted, and the digest of the
    whole equals the sum of shard digests."""
    kd, ko = acm.synth.keywords(1000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    n = 1 << 30
    text = acm.synth.device_text(n, kd, ko)
    whole = plan.scan_sorted(text)
    assert np.all(np.diff(whole["end_pos"].astype(np.int64)) >= 0)
    parts, warm = [], m.lmax - 1
    for r in range(4):
        b, e = n * r // 4, n * (r + 1) // 4
        rb = max(b - warm, 0)
        parts.append(plan.scan_sorted(text[rb:e], emit_from=b - rb, pos_base=rb))
    cat = np.concatenate(parts)
    assert cat.size == whole.size
    assert np.array_equal(cat, whole)
    assert int(plan.count(text).item()) == whole.size
    # first 64 MiB of the stream is the survey's known-answer prefix
    head = whole[whole["end_pos"] < (1 << 26)]
    assert head.size == 35453 and po.digest(head) == 0x75c631ca92f2fd08
    # the whole: the oracle's answer at full size (known_answers.json; the same as bench.py asserts)
    ka = _known_answers(2)
    assert ka["complete"] and ka["marks"][-1]["below"] == n
    assert (whole.size, po.digest(whole)) == (555000, 0xdc822ef7f043a221) == (ka["marks"][-1]["count"], int(ka["marks"][-1]["digest"], 16))


@pytest.mark.parametrize("mode", ["starts", "walk"])
def test_u32_config5_shape(torch_cuda, monkeypatch, mode):
    """BASELINE config 5 shape at test size: uint32 symbols, vocab 32,768, 10k keywords; both
    kernels for large alphabets (start-parallel, and the sparse automaton walk)."""
    if mode == "walk":
        monkeypatch.setenv("ACM_GPU_SPARSE", "walk")
    kd, ko = acm.synth.keywords(10000, sym_bytes=4)
    m, o = build_pair_packed(kd, ko, sym_size=4)
    n = 1 << 20
    text = acm.synth.text(n, kd, ko, sym_bytes=4)
    dev = acm.synth.device_text(n, kd, ko, sym_bytes=4)
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), text)
    plan = m.plan(0)
    assert plan.info.kernel == (3 if mode == "walk" else 4)   # root table in LDS either way
    want = o.scan(text)
    assert np.array_equal(plan.scan_sorted(dev), want)
    # a buffer that is not 16-byte aligned takes the CSR kernel: same records
    assert np.array_equal(plan.scan_sorted(dev[1:]), o.scan(text[1:]))
    # shards with warm-up, odd lengths
    for b, e in ((0, 1), (5, 77), (1000, 1000 + 4097), (4103, 4103 + 5000), (65536 + 7, 65536 + 7 + 100001), (n - 33, n)):
        rb = max(b - (m.lmax - 1), 0)
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=rb)
        assert np.array_equal(got, want[(want["end_pos"] >= b) & (want["end_pos"] < e)])


@pytest.mark.parametrize("mode", ["gram", "sticky"])
def test_100k_dictionary_config3_shape(torch_cuda, monkeypatch, mode):
    """BASELINE config 3's dictionary (100k keywords, 508,339 states) on 16 MiB of its text, with
    the 4-gram sieve kernel and with the sticky dense walk."""
    if mode == "sticky":
        monkeypatch.setenv("ACM_GPU_GRAM", "0")
    kd, ko = acm.synth.keywords(100000)
    m, o = build_pair_packed(kd, ko, variant=po.MEYER85)
    n = 1 << 24
    text = acm.synth.text(n, kd, ko)
    plan = m.plan(0)
    assert plan.info.kernel == (5 if mode == "gram" else 1)
    got = plan.scan_sorted(_dev(torch_cuda, text))
    cnt, dig = o.scan_mt(text, 8)
    assert got.size == cnt and po.digest(got) == dig
    want_head = o.scan(text[:1 << 20])
    assert np.array_equal(got[got["end_pos"] < (1 << 20)], want_head)


def test_segment_seams(torch_cuda, monkeypatch):
    """Buffers longer than one launch segment (2^31 symbols) are scanned segment by segment with
    an lmax-1 halo; exercised here with the segment shrunk to 64 KiB (and 4 KiB for the CSR
    kernel) so that keywords straddle many seams."""
    monkeypatch.setenv("ACM_GPU_SEGMENT_LOG2", "16")
    kd, ko = acm.synth.keywords(300)
    m, o = build_pair_packed(kd, ko)
    n = 5 * 65536 + 777
    text = acm.synth.text((n + 4095) // 4096 * 4096, kd, ko)[:n].copy()
    for seam in range(65536, n, 65536):                 # a long and a short keyword across every seam
        text[seam - 5:seam + 7] = np.frombuffer(bytes(kd[ko[7]:ko[8]]) * 3, np.uint8)[:12]
    want = o.scan(text)
    plan = m.plan(0)
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text)), want)
    cut = 3 * 65536 + 5
    ref = want[want["end_pos"] >= cut]
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text), emit_from=cut), ref)
    assert int(plan.count(_dev(torch_cuda, text)).item()) == want.size
    # CSR kernel (uint32 symbols) across seams
    monkeypatch.setenv("ACM_GPU_SEGMENT_LOG2", "12")
    kd4, ko4 = acm.synth.keywords(500, sym_bytes=4, vocab=300)
    m4, o4 = build_pair_packed(kd4, ko4, sym_size=4)
    t4 = acm.synth.text(5 * 4096, kd4, ko4, sym_bytes=4, vocab=300)[:5 * 4096 - 100]
    assert np.array_equal(m4.plan(0).scan_sorted(_dev(torch_cuda, t4)), o4.scan(t4))


def test_gram_record_chunks_carried_across_segments(torch_cuda, monkeypatch):
    """The 4-gram kernel's waves keep the chunk of record slots they are filling from one launch
    segment of a scan to the next and the holes are closed once behind the last segment: a text of
    24 segments full of matches, whole, from a cut, and into a buffer that is too small (chunks
    that begin in the buffer and end in the spill area, carried over a seam)."""
    monkeypatch.setenv("ACM_GPU_SEGMENT_LOG2", "17")
    rng = np.random.default_rng(4242)
    kws = [rng.integers(97, 104, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(12000)]
    text = rng.integers(96, 105, size=24 * (1 << 17) + 12345).astype(np.uint8)
    m, o = build_pair(kws, 1)
    plan = m.plan(0)
    assert plan.info.kernel == 5 and plan.info.records_direct == 1
    want = o.scan(text)
    assert want.size > 200000
    dev = _dev(torch_cuda, text)
    assert np.array_equal(plan.scan_sorted(dev), want)
    cut = 7 * (1 << 17) + 999
    assert np.array_equal(plan.scan_sorted(dev, emit_from=cut), want[want["end_pos"] >= cut])
    for cap in (1, 1000, want.size // 3, want.size - 1):
        rec, cnt = plan.scan(dev, capacity=cap)
        assert int(cnt.item()) == want.size and rec.shape[0] == cap
        plan.status()
    assert np.array_equal(plan.scan_sorted(dev, capacity=want.size // 3), want)      # grows and repeats
    assert np.array_equal(plan.scan_sorted(dev, capacity=want.size), want)          # exactly enough
    assert int(plan.count(dev).item()) == want.size


def _checksums(torch, rec, n):
    """order-independent fingerprints of the first n records of an int64 [cap, 2] device buffer"""
    r = rec[:n]
    pos = r[:, 0]
    length = r[:, 1] & 0xFFFFFFFF
    kw = r[:, 1] >> 32
    return (int(pos.sum().item()), int(length.sum().item()), int(kw.sum().item()),
            int((pos * 1315423911 ^ (length << 40) ^ (kw + 1)).sum().item()))


def _known_answers(config):
    """the oracle's answers at full size: tests/golden/known_answers.json, made in the build
    container by tests/golden/make_known_answers.py (count and digest of the records ending below every GiB)"""
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers.json")
    with open(path) as f:
        return json.load(f)["config%d" % config]


def _scan_whole_vs_shards(torch, plan, text, n, lmax, cap, shards=4, known=None):
    """whole scan against the union of `shards` shard scans cut by the product's own
    sharded.shard_bounds (what every rank of a multi-GPU job calls); `known`: the oracle's marks
    (known_answers.json) the whole scan's records must reproduce"""
    rec = torch.empty((cap, 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    plan.scan(text, n, records=rec, count=cnt)
    whole_n = int(cnt.item())
    assert whole_n <= cap
    if known:
        marks = [mk for mk in known["marks"] if mk["below"] <= n]
        assert marks, "no known answer inside the text"
        for mk in (marks[0], marks[len(marks) // 2], marks[-1]):
            got = acm.synth.device_digest(rec, whole_n, below=mk["below"] if mk["below"] < n else None)
            assert got == (mk["count"], int(mk["digest"], 16)), (mk, got)
    whole = _checksums(torch, rec, whole_n)
    assert int(plan.count(text, n).item()) == whole_n
    # the same text through acm_gpu_scan_ordered_device (4-gram plans: a tiled scan and ONE ordering
    # pass; the others: the bucket passes): the same record set, in canonical order
    rec2, cnt2, tmp = plan.scan_ordered(text, n, capacity=whole_n + 1000)
    assert int(cnt2.item()) == whole_n
    plan.status()
    assert _checksums(torch, rec2, whole_n) == whole
    ln = rec2[:whole_n, 1] & 0xFFFFFFFF
    ok = (rec2[1:whole_n, 0] > rec2[:whole_n - 1, 0]) | ((rec2[1:whole_n, 0] == rec2[:whole_n - 1, 0]) & (ln[1:] < ln[:-1]))
    assert bool(ok.all().item()), "acm_gpu_scan_ordered_device: records not in canonical order"
    del rec2, cnt2, tmp, ln, ok
    torch.cuda.empty_cache()
    tot, sums = 0, [0, 0, 0, 0]
    for r in range(shards):
        rb, b, e = acm.sharded.shard_bounds(n, r, shards, lmax)
        plan.scan(text[rb:e], e - rb, emit_from=b - rb, pos_base=rb, records=rec, count=cnt)
        k = int(cnt.item())
        c = _checksums(torch, rec, k)
        tot += k
        sums = [(x + y) for x, y in zip(sums, c)]
    plan.status()
    assert tot == whole_n

    def wrap(v):
        return (v + (1 << 63)) % (1 << 64) - (1 << 63)
    assert tuple(wrap(v) for v in sums) == tuple(wrap(v) for v in whole)
    return whole_n


def test_config4_sharded_eight_ranks_one_process(torch_cuda):
    """BASELINE config 4's path on one GPU: config 4's dictionary (100k keywords), its text reduced
    to 256 MiB, cut for R = 8 ranks by sharded.shard_bounds; every shard goes through the HIP scan
    with its lmax-1 warm-up (emit_from, pos_base) and canonical sort, the per-rank record sets are
    concatenated in rank order -- no global sort, exactly what gather_records does with them --
    and the result must be the oracle's record set of the whole text (count + digest), in
    canonical order.  One process, no collective."""
    torch = torch_cuda
    R, n = 8, 1 << 28
    kd, ko = acm.synth.keywords(100000)
    m, o = build_pair_packed(kd, ko, variant=po.MEYER85)
    plan = m.plan(0)
    assert plan.info.kernel == 5
    text = acm.synth.device_text(n, kd, ko)
    parts = []
    for r in range(R):
        rb, b, e = acm.sharded.shard_bounds(n, r, R, m.lmax)
        assert rb == max(b - (m.lmax - 1), 0)
        part = plan.scan_sorted(text[rb:e], emit_from=b - rb, pos_base=rb)
        assert part.size and int(part["end_pos"].min()) >= b and int(part["end_pos"].max()) < e
        parts.append(part)
    cat = np.concatenate(parts)
    host = text.cpu().numpy()
    want_n, want_digest = o.scan_mt(host, max(1, len(os.sched_getaffinity(0))))
    assert cat.size == want_n and po.digest(cat) == want_digest
    key = cat["end_pos"].astype(np.int64) * 64 + (63 - cat["length"].astype(np.int64))
    assert np.all(np.diff(key) > 0), "rank-order concatenation is not the canonical order"
    # record for record on the seam between ranks 3 and 4
    rb, b, e = acm.sharded.shard_bounds(n, 4, R, m.lmax)
    lo, hi = b - (1 << 20), b + (1 << 20)
    seam = o.scan(host[lo - 64:hi], pos_base=lo - 64, emit_from=64)
    assert np.array_equal(cat[(cat["end_pos"] >= lo) & (cat["end_pos"] < hi)], seam)


def test_multi_device_scan_c_abi_eight_shards_on_one_gpu(torch_cuda):
    """acm_gpu_multi_* (the C caller's multi-GPU entry, include/acm_gpu.h) with config 4's dictionary:
    eight shards, all of them on device 0 -- every line but the peer copy runs (shards of the root
    device take the device-to-device branch) -- against the oracle's scan of the whole text, record
    for record in canonical order; then the device-resident entry point on shards cut by
    acm_gpu_multi_shard_bounds, a too-small record buffer, a text shorter than the number of
    shards, and a single shard."""
    torch = torch_cuda
    kd, ko = acm.synth.keywords(100000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    o = po.Oracle(1, po.AC75)
    o.add_keywords_packed(kd, ko)
    n = (32 << 20) + 12345
    text = acm.synth.text((n + 4095) // 4096 * 4096, kd, ko)[:n]
    want_n, want_d = o.scan_mt(text, 8)
    mu = acm.MultiScan(m, [0] * 8)
    got = mu.scan_host(text)
    assert got.size == want_n and po.digest(got) == want_d
    assert np.all(np.diff(got["end_pos"].astype(np.int64)) >= 0)
    head = o.scan(text[:1 << 20])
    assert np.array_equal(got[:head.size], head)            # canonical order, record for record, across the first seams
    # seams: the records around every shard boundary equal the oracle's there
    for r in range(1, 8):
        rb, b, e = mu.shard_bounds(n, r)
        assert rb % 16 == 0 and rb <= b - (m.lmax - 1) and b == n * r // 8
        lo, hi = b - 4096, b + 4096
        around = o.scan(text[lo - 64:hi], pos_base=lo - 64, emit_from=64)
        sel = got[(got["end_pos"] >= lo) & (got["end_pos"] < hi)]
        assert np.array_equal(sel, around), r
    # shards already on the device
    dev = torch.from_numpy(text).cuda()
    shards = []
    for r in range(8):
        rb, b, e = mu.shard_bounds(n, r)
        shards.append(dev[rb:e])
        assert shards[-1].data_ptr() % 16 == 0
    rec = torch.empty((want_n + 5, 2), dtype=torch.int64, device="cuda")
    assert mu.scan_device(shards, n, rec) == want_n
    assert acm.synth.device_digest(rec, want_n) == (want_n, want_d)
    assert bool((rec[1:want_n, 0] >= rec[:want_n - 1, 0]).all().item())
    # a buffer that is too small: the call says how many records there are
    small = np.zeros(10, dtype=acm.RECORD_DTYPE)
    found = C.c_uint64(0)
    rc = acm.lib().acm_gpu_multi_scan_host(mu.h, text.ctypes.data, n, small.ctypes.data, 10, C.byref(found))
    assert rc == -4 and found.value == want_n
    # fewer symbols than shards; an empty text
    tiny = text[:5]
    assert np.array_equal(mu.scan_host(tiny), o.scan(tiny))
    assert mu.scan_host(text[:0]).size == 0
    # a second call on the same handle re-uses the shards' buffers (grow-only): the same records
    rec.zero_()
    assert mu.scan_device(shards, n, rec) == want_n
    assert acm.synth.device_digest(rec, want_n) == (want_n, want_d)
    mu.close()
    one = acm.MultiScan(m, [0])
    part = text[:1 << 20]
    assert np.array_equal(one.scan_host(part), o.scan(part))
    one.close()


def test_records_on_the_wire_as_eight_byte_words(torch_cuda, monkeypatch):
    """acm_gpu_wire_bits / acm_gpu_pack_records_device / acm_gpu_unpack_records_device: what a shard
    sends to the root in a multi-GPU scan.  Round trip of a scan's ordered records (bit for bit, with
    a position base beyond 2^40), the field widths of config 4's shards, and the multi-device entry
    with the wire form forced for the shards of the root device too (ACM_GPU_WIRE=2: on one GPU the
    pack -> copy -> unpack path runs for all eight shards) against the oracle."""
    torch = torch_cuda
    kd, ko = acm.synth.keywords(100000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    o = po.Oracle(1, po.AC75)
    o.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    w4 = plan.wire(1 << 34, 5 << 34)
    assert w4 is not None and w4[1:] == (34, 4)            # 2^34 positions, lengths up to 12; 17 bits of keyword id on top
    n = (16 << 20) + 777
    text = acm.synth.text((n + 4095) // 4096 * 4096, kd, ko)[:n]
    dev = torch.from_numpy(text).cuda()
    base = (1 << 41) + 4096
    rec, cnt, _ = plan.scan_ordered(dev, pos_base=base, capacity=n // 16)
    k = int(cnt.item())
    want_n, want_d = o.scan_mt(text, 8)
    assert k == want_n
    wire = plan.wire(n, base)
    assert wire == (base, 25, 4)
    packed = acm.binding.pack_records(rec[:k], wire)
    assert packed.shape == (k,) and packed.dtype == torch.int64
    back = torch.zeros((k, 2), dtype=torch.int64, device="cuda")
    acm.binding.unpack_records(packed, wire, back)
    assert torch.equal(back, rec[:k])
    assert int((packed & ((1 << 25) - 1)).max().item()) < n and bool((packed[1:] & ((1 << 25) - 1) >= packed[:-1] & ((1 << 25) - 1)).all().item())
    # fields that do not fit 64 bits: no wire form (2^60 positions + 4 + 17 bits)
    assert plan.wire(1 << 60, 0) is None
    monkeypatch.setenv("ACM_GPU_WIRE", "2")
    mu = acm.MultiScan(m, [0] * 8)
    got = mu.scan_host(text)
    assert got.size == want_n and po.digest(got) == want_d and np.all(np.diff(got["end_pos"].astype(np.int64)) >= 0)
    head = o.scan(text[:1 << 20])
    assert np.array_equal(got[:head.size], head)
    mu.close()


def _gloo_gpu_worker(rank, world, port, n, K, out_path):
    """rank of a world-2 job whose ranks share cuda:0 (a one-GPU box): product sharding, product
    HIP scan, product gather (gloo moves the records through host memory)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    kd, ko = acm.synth.keywords(K)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)

    def make_shard(read_begin, own_end):
        gb = read_begin // 4096 * 4096
        return acm.synth.device_text(own_end - gb, kd, ko, begin=gb)[read_begin - gb:]

    def scan_fn(text, emit_from, pos_base):
        rec = plan.scan_sorted(text, emit_from=emit_from, pos_base=pos_base)
        return torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.int64).reshape(-1, 2).copy()).cuda()

    # (the records of rank 1 travel as 8-byte words: packed on its device, unpacked on rank 0's)
    got = acm.sharded.scan_sharded(scan_fn, n, m.lmax, make_shard, wire_fn=plan.wire)
    if rank == 0:
        np.save(out_path, got.cpu().numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _rccl_single_rank_worker(rank, port, n, K, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    kd, ko = acm.synth.keywords(K)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)

    def make_shard(read_begin, own_end):
        return acm.synth.device_text(own_end, kd, ko)[read_begin:]

    def scan_fn(text, emit_from, pos_base):
        rec, cnt = plan.scan(text, emit_from=emit_from, pos_base=pos_base)
        k = int(cnt.item())
        plan.sort(rec, k)
        return rec[:k]

    got = acm.sharded.scan_sharded(scan_fn, n, m.lmax, make_shard)
    t = torch.ones(1, device="cuda")
    dist.all_reduce(t)                  # the timing reductions of bench.py take this road too
    np.save(out_path, got.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_entry_point_over_rccl_single_rank(torch_cuda, tmp_path):
    """The RCCL side of the sharded path as far as one GPU goes: the `nccl` process group, the
    all-gather of record counts on device tensors and a device all-reduce, with world size 1
    (peer-to-root transfers need a second GPU: the driver's 2/4/8-GPU runs)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, K = 8 << 20, 1000
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_rccl_single_rank_worker, args=(port, n, K, out), nprocs=1, join=True)
    got = np.frombuffer(np.load(out).tobytes(), dtype=acm.RECORD_DTYPE)
    kd, ko = acm.synth.keywords(K)
    o = po.Oracle(1)
    o.add_keywords_packed(kd, ko)
    assert np.array_equal(got, o.scan(acm.synth.text(n, kd, ko)))


def test_sharded_scan_two_ranks_gloo_real_hip_scan(torch_cuda, tmp_path):
    """sharded.scan_sharded end to end with the product kernel as each rank's scanner: two
    processes on this one GPU over gloo (RCCL wants one GPU per rank; the N-GPU run is the
    driver's), records gathered to rank 0 equal the oracle's scan of the whole text."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n, K = (24 << 20) + 4096 * 3 + 1234, 1000        # ragged: not a multiple of the block or the world size
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_gloo_gpu_worker, args=(2, port, n, K, out), nprocs=2, join=True)
    got = np.frombuffer(np.load(out).tobytes(), dtype=acm.RECORD_DTYPE)
    kd, ko = acm.synth.keywords(K)
    o = po.Oracle(1)
    o.add_keywords_packed(kd, ko)
    want = o.scan(acm.synth.text((n + 4095) // 4096 * 4096, kd, ko)[:n])
    assert np.array_equal(got, want)


def test_config3_full_size_properties(torch_cuda):
    """BASELINE config 3 at full size (100k keywords, 16 GiB text): the record set of the whole
    scan equals the union of four shard scans with lmax-1 warm-up (count and order-independent
    checksums), and equals the count-only entry point.  The first 16 MiB are also compared with
    the oracle in test_100k_dictionary_config3_shape."""
    torch = torch_cuda
    kd, ko = acm.synth.keywords(100000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    n = 16 << 30
    text = acm.synth.device_text(n, kd, ko)
    found = _scan_whole_vs_shards(torch, plan, text, n, m.lmax, cap=600_000_000, known=_known_answers(3))
    assert found > 400_000_000          # ~28 M matches per GiB
    del text
    torch.cuda.empty_cache()


def test_config5_full_size_properties(torch_cuda):
    """BASELINE config 5 at full size (uint32 symbols, 10k keywords, 2^30 tokens = 4 GiB)."""
    torch = torch_cuda
    kd, ko = acm.synth.keywords(10000, sym_bytes=4)
    m = acm.Machine(4)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    n = 1 << 30
    text = acm.synth.device_text(n, kd, ko, sym_bytes=4)
    found = _scan_whole_vs_shards(torch, plan, text, n, m.lmax, cap=4_000_000, known=_known_answers(5))
    assert found == 273478              # the oracle's count at full size (known_answers.json)
    del text
    torch.cuda.empty_cache()


@pytest.mark.parametrize("piece", [7, 100, 4096, 50000])
def test_streaming_pieces_equal_whole_scan(torch_cuda, novel_bytes, piece):
    """acm_gpu_stream_*: the novel fed in ragged pieces (shorter than a keyword, shorter than the
    halo, around a tile) gives the records of the whole scan; config 1 dictionary."""
    m, o = build_pair([b"he", b"she", b"his", b"hers"], 1)
    text = np.frombuffer(novel_bytes, np.uint8)[:120000 if piece < 1000 else None]
    want = o.scan(text)
    plan = m.plan(0)
    st = plan.stream(max_piece_symbols=max(piece, 16), record_capacity=want.size + 10)
    rng = np.random.default_rng(piece)
    i = 0
    while i < text.size:
        n = int(rng.integers(1, piece + 1))
        st.feed(text[i:i + n].copy())
        i += n
    got = st.finish()
    assert np.array_equal(got, want)
    st.close()
    # the plan is usable for plain scans again
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text)), want)


def test_streaming_synthetic_large_pieces(torch_cuda):
    """1k-keyword dictionary, 64 MiB fed in 8 MiB pieces from pinned host memory; digest of
    SURVEY.md App. C; then more pieces keep accumulating."""
    torch = torch_cuda
    kd, ko = acm.synth.keywords(1000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    n = 1 << 26
    host = acm.synth.device_text(n, kd, ko).cpu().pin_memory()
    st = plan.stream(max_piece_symbols=8 << 20, record_capacity=1 << 17)
    st.feed_ptr(host.data_ptr(), n)
    got = st.finish()
    assert got.size == 35453 and po.digest(got) == 0x75c631ca92f2fd08
    assert np.all(np.diff(got["end_pos"].astype(np.int64)) >= 0)
    st.close()


def test_streaming_pinned_buffers_reused_as_early_as_allowed(torch_cuda):
    """The feed contract (acm_gpu.h): a host buffer may be overwritten once the SECOND next feed
    has returned.  Three pinned buffers in rotation, each refilled right after that feed -- with
    pinned memory the copies are truly asynchronous, so a missing host-side wait would let piece
    k + 3 overwrite piece k before it has been copied."""
    torch = torch_cuda
    kd, ko = acm.synth.keywords(1000)
    m, o = build_pair_packed(kd, ko)
    piece, pieces = 4 << 20, 12
    host = acm.synth.text(piece * pieces, kd, ko)
    want = o.scan_mt(host, 8)
    plan = m.plan(0)
    st = plan.stream(max_piece_symbols=piece, record_capacity=1 << 16)
    ring = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(3)]
    for k in range(pieces):
        buf = ring[k % 3]                 # last used by feed k - 3; feeds k - 2 and k - 1 have returned
        buf.numpy()[:] = host[k * piece:(k + 1) * piece]
        st.feed_ptr(buf.data_ptr(), piece)
    got = st.finish()
    st.close()
    assert (got.size, po.digest(got)) == want
    assert np.array_equal(got[:1000], o.scan(host[:1 << 20])[:1000])


@pytest.mark.parametrize("name", ["ternary_dense", "u16_symbols"])
def test_plan_from_loaded_blob(torch_cuda, name, tmp_path):
    """A dictionary saved as a flat-table blob and loaded again (no machine, no keywords) scans
    like the machine it came from; record keyword ids resolve to spellings through the blob."""
    from aho_corasick_1975_amd import binding
    kws, text, sym = CASES[name]
    m, o = build_pair(kws, sym)
    want = o.scan(text)
    path = tmp_path / "dict.ac75"
    m.flatten().save(path)
    del m
    loaded = binding.FlatTables.load(path)
    got = loaded.plan(0).scan_sorted(_dev(torch_cuda, text))
    assert np.array_equal(got, want)
    assert want.size > 0
    for r in got[:: max(1, got.size // 50)]:
        assert loaded.keyword(int(r["keyword_id"])).size == int(r["length"])


def test_config2_plan_from_blob_digest(torch_cuda):
    from aho_corasick_1975_amd import binding
    kd, ko = acm.synth.keywords(1000)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    blob = m.flatten().to_bytes()
    plan = binding.FlatTables.from_bytes(blob).plan(0)
    got = plan.scan_sorted(acm.synth.device_text(1 << 26, kd, ko))
    assert got.size == 35453 and po.digest(got) == 0x75c631ca92f2fd08


def _fn_ptr(kat, name):
    import ctypes as C
    return C.cast(getattr(kat, name), C.c_void_p)


def test_case_insensitive_comparator_on_the_gpu(torch_cuda, kat, novel_bytes):
    """Comparator classes: the reference's case-insensitive matching
    (generic_test.c:48-54) with byte symbols -- the device maps the text through the class table,
    then walks it with the dense kernel; records equal the oracle's with the same comparator."""
    kws = [b"He", b"SHE", b"his", b"hErs"]
    m = acm.Machine(1, cmp=_fn_ptr(kat, "kat_casecmp8"))
    o = po.Oracle(1, po.MEYER85, cmp=_fn_ptr(kat, "kat_casecmp8"))
    for kw in kws:
        m.add_keyword(kw)
        o.add_keyword(kw)
    plan = m.plan_classes(0)
    assert plan.info.kernel == 1
    want = o.scan(novel_bytes)
    assert want.size > 11676
    dev = _dev(torch_cuda, novel_bytes)
    assert np.array_equal(plan.scan_sorted(dev), want)
    assert int(plan.count(dev).item()) == want.size
    # unaligned buffers, odd lengths, shards with warm-up
    for b, e in ((1, 7), (3, 100001), (77, 12345), (len(novel_bytes) - 9, len(novel_bytes))):
        rb = max(b - 3, 0)
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=rb)
        assert np.array_equal(got, want[(want["end_pos"] >= b) & (want["end_pos"] < e)])
    # streaming pieces go through the same mapping
    st = plan.stream(max_piece_symbols=50000, record_capacity=1 << 16)
    host = np.frombuffer(novel_bytes, np.uint8)
    for off in range(0, host.size, 50000):
        st.feed(host[off:off + 50000])
    assert np.array_equal(st.finish(), want)
    st.close()


def test_interleaved_classes_and_u16_on_the_gpu(torch_cuda, kat):
    import ctypes as C
    rng = np.random.default_rng(5)
    kws = [bytes(rng.integers(0, 256, size=rng.integers(1, 6)).astype(np.uint8)) for _ in range(60)]
    m = acm.Machine(1, cmp=_fn_ptr(kat, "kat_mod7cmp8"))
    o = po.Oracle(1, po.MEYER85, cmp=_fn_ptr(kat, "kat_mod7cmp8"))
    for kw in kws:
        m.add_keyword(kw)
        o.add_keyword(kw)
    text = rng.integers(0, 256, size=300001).astype(np.uint8)
    plan = m.plan_classes(0)
    assert np.array_equal(plan.scan_sorted(_dev(torch_cuda, text)), o.scan(text))
    # 2-byte symbols, case-insensitive over UTF-16 units
    kws16 = [np.array([ord(c) for c in w], np.uint16) for w in ("Été", "STRASSE", "naïve", "Ωmega", "x")]
    m16 = acm.Machine(2, cmp=_fn_ptr(kat, "kat_casecmp16"))
    o16 = po.Oracle(2, po.MEYER85, cmp=_fn_ptr(kat, "kat_casecmp16"))
    for kw in kws16:
        m16.add_keyword(kw)
        o16.add_keyword(kw)
    text16 = np.array([ord(c) for c in "l'été ÉTÉ Strasse straße NAÏVE ωMEGA ΩMEGa xX " * 999], np.uint16)
    plan16 = m16.plan_classes(0)
    want16 = o16.scan(text16)
    assert want16.size >= 999 * 8
    assert np.array_equal(plan16.scan_sorted(_dev(torch_cuda, text16)), want16)
    assert np.array_equal(plan16.scan_sorted(_dev(torch_cuda, text16)[1:]), o16.scan(text16[1:]))


GENERIC_TEST_1_KEYWORDS = ["he", "she", "sheers", "his", "hi", "hers", "ushers", "abcde", "bcd", "hers", "hen", "hen", "bcdef", "pen",
                           "cdefg", "pen", "bcd", "abc", "abcd", "abcde", "bcde", "cde", "cd", "bc", "u", "uu"]
GENERIC_TEST_1_TEXT = "He found his pencil, but she could not find hers (Hi! Ushers !! --abcdefgh--)"


def _u32(s):
    return np.array([ord(c) for c in s], np.uint32)


def test_reference_generic_test_1_wchar_alphacmp_on_the_gpu(torch_cuda, kat, novel_bytes):
    """The reference's flagship "any ordered alphabet" example on the GPU: generic_test.c:62-164,
    4-byte wchar_t symbols under the case-insensitive alphacmp (:48-54).  The 2^32 symbol values
    cannot be enumerated: the plan holds the classes of the dictionary's own symbols and
    classifies the symbols of a text with the machine's comparator when it first meets them
    (acm_flatten_classes with sym_bytes 4).  Records equal the oracle's with the same comparator."""
    import ctypes as C
    kat.setlocale_utf8()
    cmp32 = _fn_ptr(kat, "kat_casecmp32")
    m = acm.Machine(4, cmp=cmp32)
    o = po.Oracle(4, po.MEYER85, cmp=cmp32)
    for w in GENERIC_TEST_1_KEYWORDS:
        m.add_keyword(_u32(w))
        o.add_keyword(_u32(w))
    assert m.nb_keywords == 21
    plan = m.plan_classes(0)
    text = _u32(GENERIC_TEST_1_TEXT)
    want = o.scan(text)
    assert want.size == 27                  # he, u, hi, his, pen, ...
    got = plan.scan_sorted(_dev(torch_cuda, text))
    assert np.array_equal(got, want)
    assert int(plan.count(_dev(torch_cuda, text)).item()) == 27
    # MatchHolder spelling is the dictionary's, not the text's (generic_test.c output {'he'} for "He")
    word, _ = m.keyword(int(got[0]["keyword_id"]))
    assert word == tuple(ord(c) for c in "he") and int(got[0]["end_pos"]) == 1
    # a long text with symbols the dictionary never saw (upper case, punctuation, Latin-1, beyond the BMP)
    novel = np.frombuffer(novel_bytes, np.uint8).astype(np.uint32)
    novel[::997] = 0x1F600 + (np.arange(novel[::997].size) % 50)
    novel[5::1013] = ord("É")
    want = o.scan(novel)
    dev = _dev(torch_cuda, novel)
    assert want.size > 20000
    assert np.array_equal(plan.scan_sorted(dev), want)
    assert np.array_equal(plan.scan_sorted(dev), want)      # nothing new to classify the second time
    for b, e in ((1, 7), (3, 100001), (77, 12345), (novel.size - 9, novel.size)):
        rb = max(b - 5, 0)
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=rb)
        assert np.array_equal(got, want[(want["end_pos"] >= b) & (want["end_pos"] < e)])
    # the dictionary grows: new keywords (new symbols among them) go into the delta plan
    for w in ("Mrs", "DALLOWAY", "Ébloui", "she said"):
        m.add_keyword(_u32(w))
        o.add_keyword(_u32(w))
    plan.update(m)
    want = o.scan(novel)
    assert np.array_equal(plan.scan_sorted(dev), want)
    # a comparator that is no order is refused
    bad = acm.Machine(4, cmp=_fn_ptr(kat, "kat_cyclic_cmp32"))
    for w in ("abc", "bca", "cab"):
        bad.add_keyword(_u32(w))
    with pytest.raises(acm.binding.ACMError):
        bad.plan_classes(0)


def test_comparator_classes_text_of_two_million_distinct_symbols(torch_cuda, kat):
    """4-byte symbols under a comparator, a text that brings 2^21 symbols the dictionary never saw:
    every one of them is listed once for the host to classify.  The list of one pass is bounded, so
    the scan takes several passes -- the list doubles when a pass fills it, and a full list stops
    claiming table slots (a table overrun by claims used to turn every later miss into a walk over
    all of it: the scan looked hung)."""
    kat.setlocale_utf8()
    cmp32 = _fn_ptr(kat, "kat_casecmp32")
    m = acm.Machine(4, cmp=cmp32)
    o = po.Oracle(4, po.MEYER85, cmp=cmp32)
    for w in ("he", "she", "his", "hers"):
        m.add_keyword(_u32(w))
        o.add_keyword(_u32(w))
    plan = m.plan_classes(0)
    n = 1 << 21
    text = (0x10000 + np.arange(n, dtype=np.uint32) * 3).astype(np.uint32)      # 2 M distinct symbols, none of the dictionary's
    for at, w in ((5, "uSHErs"), (100000, "His"), (n - 4, "hers")):
        text[at:at + len(w)] = _u32(w)
    want = o.scan(text)
    assert want.size == 6
    dev = _dev(torch_cuda, text)
    assert np.array_equal(plan.scan_sorted(dev), want)
    assert np.array_equal(plan.scan_sorted(dev), want)      # nothing new the second time: one pass


@pytest.mark.parametrize("seed", range(int(os.environ.get("ACM_SOAK_SEEDS", "1"))))
def test_incremental_updates_of_a_start_parallel_plan(torch_cuda, seed):
    """keywords added while the plan is in use (reference README.md:352-356,
    generic_test.c:214-229).  uint32 symbols: the plan is edited in place (a few table words per
    keyword, written in front of the next scan); after every batch the records equal the oracle's."""
    rng = np.random.default_rng(11 + seed)
    V = 5000 if seed == 0 else int(rng.integers(30, 40000))
    def word(lo, hi):
        return rng.integers(0, V, size=rng.integers(lo, hi)).astype(np.uint32)
    base = [word(2, 6) for _ in range(400)]
    m, o = build_pair(base, 4)
    text = rng.integers(0, V, size=200000).astype(np.uint32)
    # plant occurrences of keywords that exist now and of some that come later
    later = [word(1, 7) for _ in range(300)]
    later += [np.concatenate([base[i], word(1, 3)]) for i in range(40)]            # extensions of keywords
    later += [base[i][:max(1, base[i].size - 1)] for i in range(40, 80)]            # prefixes: inner states turn terminal
    later += [np.array([V + 7, 3, 4], np.uint32), np.array([V + 7], np.uint32), np.array([12, V + 900], np.uint32)]  # symbols beyond the root table
    later += [np.array([base[0][0]], np.uint32)]                                    # a single symbol that is already a root child
    for i, kw in enumerate(base[:200] + later):
        at = int(rng.integers(0, text.size - 16))
        text[at:at + kw.size] = kw
    dev = _dev(torch_cuda, text)
    plan = m.plan(0)
    assert plan.info.kernel == 4
    assert np.array_equal(plan.scan_sorted(dev), o.scan(text))
    order = rng.permutation(len(later))
    step = 0
    for lo in range(0, len(later), 37):
        for j in order[lo:lo + 37]:
            m.add_keyword(later[j])
            o.add_keyword(later[j])
        plan.update(m)
        want = o.scan(text)
        got = plan.scan_sorted(dev)
        assert got.size == want.size and np.array_equal(got, want), "after batch %d" % step
        if step % 3 == 0:
            assert int(plan.count(dev).item()) == want.size
            assert np.array_equal(plan.scan_sorted(dev[1:]), o.scan(text[1:]))      # unaligned buffer: aligned copy inside
        if step % 4 == 1:                                                            # and a stream on the edited plan
            st = plan.stream(max_piece_symbols=50000, record_capacity=want.size + 16)
            for off in range(0, text.size, 50000):
                st.feed(text[off:off + 50000])
            assert np.array_equal(st.finish(), want)
            st.close()
        step += 1
    # the edited plan equals one built from scratch
    assert np.array_equal(m.plan(0).scan_sorted(dev), plan.scan_sorted(dev))
    # one keyword at a time through acm_scan (machine-cached plan)
    for kw in [word(2, 5) for _ in range(25)]:
        m.add_keyword(kw)
        o.add_keyword(kw)
        assert np.array_equal(m.scan_host(text[:50000]), o.scan(text[:50000]))


def test_plan_update_rebuilds_dense_and_class_plans(torch_cuda, kat):
    m, o = build_pair([b"he", b"she"], 1)
    text = b"ushers and heroes; she sells hers and HIS " * 300
    plan = m.plan(0)
    dev = _dev(torch_cuda, text)
    assert np.array_equal(plan.scan_sorted(dev), o.scan(text))
    for w in (b"his", b"hers", b"s", b"sells hers"):
        m.add_keyword(w)
        o.add_keyword(w)
        plan.update(m)
        assert np.array_equal(plan.scan_sorted(dev), o.scan(text))
    mc = acm.Machine(1, cmp=_fn_ptr(kat, "kat_casecmp8"))
    oc = po.Oracle(1, po.MEYER85, cmp=_fn_ptr(kat, "kat_casecmp8"))
    for w in (b"He", b"she"):
        mc.add_keyword(w)
        oc.add_keyword(w)
    pc = mc.plan_classes(0)
    assert np.array_equal(pc.scan_sorted(dev), oc.scan(text))
    for w in (b"HIS", b"hErs"):
        mc.add_keyword(w)
        oc.add_keyword(w)
    pc.update(mc)
    assert np.array_equal(pc.scan_sorted(dev), oc.scan(text))


def test_gram_kernel_small_alphabet_dense_matches(torch_cuda):
    """4-gram sieve kernel on a dictionary built to stress it: 6-letter alphabet (almost every
    4-gram is a keyword prefix), 20k keywords of 4..9 symbols, text with symbols outside the
    alphabet, matches everywhere; shards, odd lengths and unaligned buffers included."""
    rng = np.random.default_rng(21)
    kws = [bytes(rng.integers(ord("a"), ord("g"), size=rng.integers(4, 10)).astype(np.uint8)) for _ in range(20000)]
    m, o = build_pair(kws, 1)
    text = rng.integers(ord("a") - 1, ord("h"), size=300007).astype(np.uint8)
    plan = m.plan(0)
    assert plan.info.kernel == 5
    want = o.scan(text)
    assert want.size > 50000
    dev = _dev(torch_cuda, text)
    got = plan.scan_sorted(dev)
    assert got.size == want.size and np.array_equal(got, want)
    assert int(plan.count(dev).item()) == want.size
    for b, e in ((0, 3), (0, 4), (1, 9), (5, 4099), (1000, 200001), (text.size - 5, text.size)):
        rb = max(b - (m.lmax - 1), 0)
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=rb)
        assert np.array_equal(got, want[(want["end_pos"] >= b) & (want["end_pos"] < e)]), (b, e)


def test_gram_kernel_every_position_sends_a_walk(torch_cuda):
    """The 4-gram kernel's queues at their limits: all rotations of a 10-letter string are keywords
    and the text is that string over and over, so every position starts a keyword, every lane of
    every batch passes both sieves, and 64 walk candidates arrive at a time (a walk queue of 96 has
    to make room first); 12k random keywords of 4-9 letters beside them force the kernel."""
    rng = np.random.default_rng(77)
    base = np.frombuffer(b"abcdefghij", np.uint8)
    kws = [np.roll(base, -k).copy() for k in range(base.size)]
    kws += [rng.integers(97, 123, size=rng.integers(4, 10)).astype(np.uint8) for _ in range(12000)]
    text = np.tile(base, 120000)                               # (a match per position: more than a wave's region of the item buffer holds)
    text[150000:150100] = rng.integers(97, 123, size=100)      # a break in the pattern
    m, o = build_pair(kws, 1)
    plan = m.plan(0)
    assert plan.info.kernel == 5 and plan.info.dense_rows > 32768
    want = o.scan(text)
    assert want.size > text.size - 200
    dev = _dev(torch_cuda, text)
    got = plan.scan_sorted(dev)
    assert got.size == want.size and np.array_equal(got, want)
    assert int(plan.count(dev).item()) == want.size


def _random_case(rng, kind):
    """Random dictionary + text meant for one kernel family; returns (keywords, text, sym_bytes, env)."""
    if kind == "dense":            # byte alphabet, few states: continuation-mode dense kernel
        lo = int(rng.integers(0, 200)); span = int(rng.integers(1, 40))
        kws = [rng.integers(lo, lo + span, size=rng.integers(1, 14)).astype(np.uint8) for _ in range(int(rng.integers(1, 300)))]
        text = rng.integers(max(lo - 2, 0), min(lo + span + 2, 256), size=int(rng.integers(1, 200000))).astype(np.uint8)
        return kws, text, 1, {}
    if kind in ("gram", "gramold", "sticky"):  # small alphabet, > 32768 states, keywords >= 4 symbols
        span = int(rng.integers(5, 10))
        kws = [rng.integers(97, 97 + span, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(int(rng.integers(11000, 14000)))]
        text = rng.integers(96, 97 + span + 1, size=int(rng.integers(50000, 400000))).astype(np.uint8)
        # gram: scan_gram2_kernel (lane-local sieve, the default for such dictionaries); gramold: scan_gram_kernel
        return kws, text, 1, ({"ACM_GPU_GRAM": "0"} if kind == "sticky" else ({"ACM_GPU_GRAM2": "0"} if kind == "gramold" else {}))
    if kind == "gram26":            # the widest alphabet scan_gram2_kernel takes (26 symbols + "other": LDS filled to the last KiB)
        kws = [rng.integers(97, 123, size=rng.integers(4, 12)).astype(np.uint8) for _ in range(int(rng.integers(9000, 11000)))]
        text = rng.integers(96, 124, size=int(rng.integers(50000, 400000))).astype(np.uint8)
        for _ in range(4000):       # keyword heads of 3 to 7 symbols: every stage sees hits and near misses
            w = kws[int(rng.integers(0, len(kws)))]
            k = min(w.size, 3 + int(rng.integers(0, 5)))
            at = int(rng.integers(0, text.size - 12))
            text[at:at + k] = w[:k]
        return kws, text, 1, {}
    if kind == "gram30":            # the widest alphabet the exact 4-gram index takes (28-30 classes), Bloom filters on
        lo = int(rng.integers(0, 200)); span = int(rng.integers(27, 30))
        kws = [rng.integers(lo, lo + span, size=rng.integers(4, 10)).astype(np.uint8) for _ in range(int(rng.integers(11000, 13000)))]
        text = rng.integers(max(lo - 1, 0), lo + span + 1, size=int(rng.integers(50000, 300000))).astype(np.uint8)
        for _ in range(3000):       # keyword heads of 4 to 6 symbols: the filters and the rank tables see hits and near misses
            w = kws[int(rng.integers(0, len(kws)))]
            k = min(w.size, 4 + int(rng.integers(0, 3)))
            at = int(rng.integers(0, text.size - 12))
            text[at:at + k] = w[:k]
        return kws, text, 1, {}
    if kind == "gramsmall":         # the 4-gram kernel forced onto small dictionaries (few 4-grams set, lmax down to 4)
        lo = int(rng.integers(0, 220)); span = int(rng.integers(1, 29))
        top = int(rng.integers(5, 12))
        shortest = 4 if rng.integers(0, 2) else 1      # with or without keywords of 1-3 symbols
        kws = [rng.integers(lo, lo + span, size=rng.integers(shortest, top)).astype(np.uint8) for _ in range(int(rng.integers(1, 1500)))]
        kws.append(rng.integers(lo, lo + span, size=top).astype(np.uint8))      # at least one keyword of 4 symbols or more
        text = rng.integers(max(lo - 1, 0), min(lo + span + 1, 256), size=int(rng.integers(1, 200000))).astype(np.uint8)
        return kws, text, 1, {"ACM_GPU_GRAM": "2"}
    if kind == "gramheads":         # narrow alphabet, small dictionary of keywords of 4 symbols or more, the text full of keyword heads: 4-gram kernel
        lo = int(rng.integers(0, 220)); span = int(rng.integers(16, 29))
        kws = [rng.integers(lo, lo + span, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(int(rng.integers(1, 400)))]
        text = rng.integers(max(lo - 1, 0), min(lo + span + 1, 256), size=int(rng.integers(1, 300000))).astype(np.uint8)
        # dense candidates: re-use keyword heads of 3 to 6 symbols so that every stage sees rejects
        for _ in range(min(3000, text.size // 16)):
            w = kws[int(rng.integers(0, len(kws)))]
            k = min(w.size, 3 + int(rng.integers(0, 4)))
            at = int(rng.integers(0, max(text.size - 12, 1)))
            if at + k <= text.size:
                text[at:at + k] = w[:k]
        return kws, text, 1, {"ACM_GPU_GRAM": "2"}
    if kind in ("wide", "wideshort"):   # more than 29 symbols in use, > 32768 states: hashed 4-byte windows (and shorter ones)
        lo = int(rng.integers(0, 120)); span = int(rng.integers(31, 136))
        if rng.integers(0, 3) == 0:
            lo, span = 0, 256           # every byte value in use: dense rows of width 256
        kws = [rng.integers(lo, lo + span, size=rng.integers(4 if kind == "wide" else 1, 11)).astype(np.uint8) for _ in range(int(rng.integers(9000, 12000)))]
        text = rng.integers(lo, lo + span, size=int(rng.integers(50000, 400000))).astype(np.uint8)
        # dense 4-gram hits: re-use keyword heads so that the table, not only the Bloom bits, is exercised
        for _ in range(3000):
            w = kws[int(rng.integers(0, len(kws)))]
            k = min(w.size, 4 + int(rng.integers(0, 3)))
            at = int(rng.integers(0, text.size - 12))
            text[at:at + k] = w[:k]
        return kws, text, 1, {}
    if kind == "short":             # > 32768 states and keywords of 1-3 symbols too: the 4-gram kernel with its short-keyword path
        kws = [rng.integers(97, 123, size=rng.integers(1, 12)).astype(np.uint8) for _ in range(int(rng.integers(10000, 12000)))]
        text = rng.integers(97, 123, size=int(rng.integers(50000, 300000))).astype(np.uint8)
        return kws, text, 1, {}
    if kind == "shortlds":          # the same with few keywords of 1 and 2 symbols: scan_short_kernel has the keyword ids in LDS
        kws = [rng.integers(97, 123, size=rng.integers(3, 12)).astype(np.uint8) for _ in range(int(rng.integers(10000, 12000)))]
        kws += [rng.integers(97, 123, size=1).astype(np.uint8) for _ in range(2)] + [rng.integers(97, 123, size=2).astype(np.uint8) for _ in range(20)]
        text = rng.integers(97, 123, size=int(rng.integers(50000, 300000))).astype(np.uint8)
        return kws, text, 1, {}
    sym = 2 if kind.endswith("16") else 4
    dt = np.uint16 if sym == 2 else np.uint32
    V = int(rng.integers(5, 3000))
    kws = [rng.integers(0, V, size=rng.integers(1, 9)).astype(dt) for _ in range(int(rng.integers(1, 2000)))]
    text = rng.integers(0, V + 3, size=int(rng.integers(1, 200000))).astype(dt)
    return kws, text, sym, ({"ACM_GPU_SPARSE": "walk"} if kind.startswith("walk") else {})


def test_plan_timing_samples_every_nth_launch(torch_cuda):
    """acm_gpu_plan_timing (plan, N): HIP events around every N-th launch only (what bench.py uses: the
    events of a launch cost about 10 us on the stream).  Seven scans at N = 3: launches 0, 3 and 6."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    kws = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(300)]
    text = rng.integers(97, 123, size=1 << 20, dtype=np.uint8)
    m, o = build_pair(kws, 1)
    plan = m.plan(0)
    dev = _dev(torch, text)
    want = o.count(text)
    for every, scans, sampled in ((1, 4, 4), (3, 7, 3), (5, 5, 1), (2, 1, 1)):
        plan.timing(every)
        for _ in range(scans):
            rec, cnt = plan.scan(dev, capacity=want + 8)
        torch.cuda.synchronize()
        scan_ms, all_ms, launches = plan.timing_read_all()
        plan.timing(False)
        assert launches == sampled, (every, scans, launches)
        assert 0 < scan_ms <= all_ms and int(cnt.item()) == want
    plan.timing(False)
    plan.scan(dev, capacity=want + 8)
    assert plan.timing_read_all()[2] == 0


def test_comm_gather_world_of_one_over_the_real_rccl(torch_cuda):
    """acm_gpu_comm_*: the C ABI's gather of the ranks' records over RCCL, with librccl.so itself, at
    the one world size a one-GPU box allows: the communicator is made through the library's own
    entry points (unique id, init rank), the counts go through ncclAllGather, the root's own
    records through the device copy; a too-small buffer is ACM_GPU_E_OVERFLOW with the total."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    kws = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(400)]
    text = rng.integers(97, 123, size=1 << 20, dtype=np.uint8)
    m, o = build_pair(kws, 1)
    want = o.scan(text)
    plan = m.plan(0)
    rec, cnt, _ = plan.scan_ordered(_dev(torch, text), capacity=want.size + 8)
    n = int(cnt.item())
    assert n == want.size
    comm = acm.Comm(acm.Comm.unique_id(), 0, 1)
    out = torch.zeros((want.size, 2), dtype=torch.int64, device="cuda")
    total, counts = comm.gather_records(plan, rec, n, 0, text.size, out)
    torch.cuda.synchronize()
    assert total == want.size and counts == [want.size]
    assert np.array_equal(np.frombuffer(out.cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE), want)
    with pytest.raises(acm.ACMError) as ei:
        comm.gather_records(plan, rec, n, 0, text.size, out[:want.size - 1])
    assert ei.value.code == -4
    total, counts = comm.gather_records(None, rec, 0, 0, text.size, out)    # a rank with nothing to report
    assert total == 0 and counts == [0]
    comm.close()


@pytest.mark.parametrize("world,root,wire", [(2, 0, 1), (3, 2, 1), (3, 0, 0), (5, 1, 1)])
def test_comm_gather_ranks_as_threads_over_the_loopback_transport(torch_cuda, world, root, wire):
    """The same gather at world sizes of 2, 3 and 5 on ONE GPU: the ranks are threads of a worker
    process (tests/comm_loopback_worker.py), librccl.so is replaced by a loopback transport with the
    same entry points (tests/helpers/loopback_comm.hip, ACM_GPU_COMM_LIB) -- who sends what to whom,
    the offsets on the root, the 8-byte wire form and the 16-byte one, a root that is not rank 0, the
    too-small root buffer reported to every rank: against the oracle's records of the whole text.
    (RCCL's own send / receive between GPUs is what only a multi-GPU node can run.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "helpers", "libloopback_comm.so")
    src = os.path.join(here, "helpers", "loopback_comm.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src, "-pthread"], check=True)
    env = dict(os.environ, ACM_GPU_COMM_LIB=so)
    r = subprocess.run([sys.executable, os.path.join(here, "comm_loopback_worker.py"), str(world), str(root), str(wire)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK world=%d" % world in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("kind", ["gram", "gramold", "short", "shortlds"])
@pytest.mark.parametrize("seg_log2", [0, 14])
@pytest.mark.parametrize("close_sort", ["counted", "network"])
def test_record_chunks_of_4096_slots(torch_cuda, monkeypatch, kind, seg_log2, close_sort):
    """Scans of 64 MiB and more reserve their records 4,096 slots at a time instead of 1,024
    (EmitCtx::rec_chunk; every wave adds to the same counter).  ACM_GPU_REC_CHUNK=4096 selects the
    big chunks for a small text (and, with the small ones, close_holes_kernel's other way to sort
    its descriptors): whole scans, a buffer that is too small (the chunks past its
    capacity go to the spill area and come back into the holes), shards, launch segments of 16 Ki
    symbols (a wave carries its open chunk from one segment to the next), against the oracle."""
    rng = np.random.default_rng(77 + seg_log2 + sum(map(ord, kind)))
    kws, text, sym, env = _random_case(rng, kind)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("ACM_GPU_REC_CHUNK", "4096" if close_sort == "counted" else "1024")
    if close_sort == "network":     # close_holes_kernel's fallback for crowded buckets: the bitonic network
        monkeypatch.setenv("ACM_GPU_CLOSE_SORT", "network")
    if seg_log2:
        monkeypatch.setenv("ACM_GPU_SEGMENT_LOG2", str(seg_log2))
    m, o = build_pair(kws, sym)
    plan = m.plan(0)
    assert plan.info.kernel == 5 and plan.info.records_direct == 1
    want = o.scan(text)
    assert want.size > 1000
    dev = _dev(torch_cuda, text)
    assert np.array_equal(plan.scan_sorted(dev), want)
    rec, cnt = plan.scan(dev, capacity=want.size // 3)
    assert int(cnt.item()) == want.size and rec.shape[0] == want.size // 3
    assert np.array_equal(plan.scan_sorted(dev, capacity=want.size // 3), want)
    rec, cnt = plan.scan(dev, capacity=want.size)      # exactly as many slots as records: every hole closed
    assert int(cnt.item()) == want.size
    got = np.frombuffer(rec[:want.size].cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE)
    assert np.array_equal(np.sort(got, order=("end_pos", "length", "keyword_id")), np.sort(want, order=("end_pos", "length", "keyword_id")))
    lo = text.size // 3
    part = plan.scan_sorted(dev[lo - 16:], emit_from=16, pos_base=lo - 16) if lo >= 16 else None
    if part is not None:    # (a shard that starts 16 symbols early: matches that end in it, global positions)
        o_part = want[want["end_pos"] >= lo]
        maxlen = max(len(k) for k in kws)
        if maxlen <= 17:
            assert np.array_equal(part, o_part)


@pytest.mark.parametrize("kind,seed", [(k, s) for k in ("dense", "gramheads", "gram", "gramold", "gram26", "gram30", "gramsmall", "wide", "wideshort", "sticky", "short", "shortlds", "starts16", "starts32", "walk16", "walk32")
                                       for s in range(int(os.environ.get("ACM_SOAK_SEEDS", "3")))])   # ACM_SOAK_SEEDS=14: a soak run
def test_randomized_dictionaries_and_shards(torch_cuda, monkeypatch, kind, seed):
    """Random dictionaries and texts through every kernel family; whole scans, count-only scans and
    random shards (emit_from, pos_base, odd offsets and lengths) against the oracle."""
    rng = np.random.default_rng(1000 * seed + sum(map(ord, kind)))
    kws, text, sym, env = _random_case(rng, kind)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if seed == 2:   # launch segments of 8 Ki symbols instead of 2^31: every kernel family across many seams
        monkeypatch.setenv("ACM_GPU_SEGMENT_LOG2", "13")
    if seed == 1:   # the 4-gram kernel's peek entries in their 8-byte form (automata of 8 M states and more)
        monkeypatch.setenv("ACM_GPU_PEEK8", "1")
    # plant some keywords so that long matches exist
    for _ in range(min(200, text.size // 50)):
        w = kws[int(rng.integers(0, len(kws)))]
        if w.size < text.size:
            at = int(rng.integers(0, text.size - w.size))
            text[at:at + w.size] = w
    m, o = build_pair(kws, sym)
    plan = m.plan(0)
    expect = {"dense": 1, "gramheads": 5, "gram": 5, "gramold": 5, "gram26": 5, "gram30": 5, "gramsmall": 5, "wide": 5, "wideshort": 5, "sticky": 1, "short": 5, "shortlds": 5, "starts16": 4, "starts32": 4, "walk16": 3, "walk32": 3}[kind]
    if kind in ("gram", "gramold", "gram26", "gram30", "wide", "wideshort", "sticky", "short", "shortlds"):
        assert plan.info.dense_rows > 32768, "the generator is meant to give more states than the LDS scheme takes"
    if kind in ("gram", "gramold", "gram26", "gram30"):
        assert plan.info.variant == (2 if kind in ("gram", "gram26") else 0), (kind, plan.info.variant)
    if kind in ("short", "shortlds"):   # keywords of 1-3 symbols: a pass of their own (scan_short_kernel) behind the 4-gram kernel;
        # every letter and most pairs are keywords: 38 K ids, read from HBM (8) -- a few thousand: in LDS
        assert plan.info.variant == 2 | 4 | (8 if kind == "short" else 0), plan.info.variant
    if kind == "dense":     # small dictionaries whose hot rows outgrow LDS go to the 4-gram kernel by themselves
        assert plan.info.kernel in (1, 5, 6), plan.info.kernel
    else:
        assert plan.info.kernel == expect, (kind, plan.info.kernel)
    want = o.scan(text)
    dev = _dev(torch_cuda, text)
    got = plan.scan_sorted(dev)
    assert got.size == want.size and np.array_equal(got, want)
    assert int(plan.count(dev).item()) == want.size
    if want.size > 7:   # a record buffer that is too small: the total is still reported, nothing written past the end
        rec, cnt = plan.scan(dev, capacity=7)
        assert int(cnt.item()) == want.size and rec.shape[0] == 7
        assert np.array_equal(plan.scan_sorted(dev, capacity=7), want)      # grows and repeats
    # host buffers in, records out: acm_scan on the machine (its own cached plan) and on the plan
    head = text[:min(text.size, 40000)]
    want_head = want[want["end_pos"] < head.size]
    assert np.array_equal(m.scan_host(head), want_head)
    assert np.array_equal(plan.scan_host(head), want_head)
    # the same text fed to a stream in pieces of random sizes
    piece_max = int(rng.integers(1000, 60000))
    stream = plan.stream(max_piece_symbols=piece_max, record_capacity=max(want.size, 1) + 16)
    off = 0
    while off < text.size:
        step = int(rng.integers(1, piece_max + 1))
        stream.feed(text[off:off + step])
        off += step
    got = stream.finish()
    stream.close()
    assert got.size == want.size and np.array_equal(got, want)
    lmax = m.lmax
    for _ in range(6):
        b = int(rng.integers(0, text.size))
        e = int(rng.integers(b, min(text.size, b + 70000))) + 1
        e = min(e, text.size)
        rb = max(b - (lmax - 1), 0)
        base = int(rng.integers(0, 1 << 40))
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=base)
        sel = want[(want["end_pos"] >= b) & (want["end_pos"] < e)].copy()
        sel["end_pos"] += base - rb
        assert np.array_equal(got, sel), (kind, seed, b, e)


@pytest.mark.parametrize("sym", [1, 4])
def test_plan_of_an_empty_machine_and_its_first_keywords(torch_cuda, sym):
    """A plan made before any keyword exists finds nothing (generic_test.c:70 asserts the same of
    acm_match); acm_gpu_plan_update then takes it through its first keywords."""
    dt = {1: np.uint8, 4: np.uint32}[sym]
    m, o = build_pair([], sym)
    text = (np.arange(5000) % 7 + 1).astype(dt)
    dev = _dev(torch_cuda, text)
    plan = m.plan(0)
    assert plan.scan_sorted(dev).size == 0 and int(plan.count(dev).item()) == 0
    for kw in ([3], [1, 2, 3], [7, 1], [2, 3, 4, 5, 6], [5]):
        m.add_keyword(np.array(kw, dt))
        o.add_keyword(np.array(kw, dt))
        plan.update(m)
        want = o.scan(text)
        assert want.size > 0 and np.array_equal(plan.scan_sorted(dev), want)


def test_eight_byte_symbols_on_the_gpu(torch_cuda):
    """ACM_CMP_DEFAULT over 8-byte symbols (sizes 1, 2, 4, 8): the device interns the
    text through a hash table of the dictionary's symbols and walks 4-byte ids."""
    rng = np.random.default_rng(8)
    vocab = rng.integers(0, 1 << 63, size=3000, dtype=np.uint64)
    kws = [vocab[rng.integers(0, vocab.size, size=rng.integers(1, 7))] for _ in range(1500)]
    m, o = build_pair(kws, 8)
    text = vocab[rng.integers(0, vocab.size, size=200003)]
    noise = rng.integers(0, text.size, size=20000)
    text[noise] = rng.integers(0, 1 << 63, size=noise.size, dtype=np.uint64)      # symbols the dictionary has never seen
    for _ in range(300):
        w = kws[int(rng.integers(0, len(kws)))]
        at = int(rng.integers(0, text.size - w.size))
        text[at:at + w.size] = w
    plan = m.plan(0)
    assert plan.info.kernel == 4
    want = o.scan(text)
    assert want.size > 300
    dev = _dev(torch_cuda, text)
    got = plan.scan_sorted(dev)
    assert got.size == want.size and np.array_equal(got, want)
    assert int(plan.count(dev).item()) == want.size
    for b, e in ((0, 1), (3, 70001), (100000, text.size)):
        rb = max(b - (m.lmax - 1), 0)
        got = plan.scan_sorted(dev[rb:e], emit_from=b - rb, pos_base=rb)
        assert np.array_equal(got, want[(want["end_pos"] >= b) & (want["end_pos"] < e)])
    # host-buffer path and streaming take 8-byte symbols too
    assert np.array_equal(plan.scan_host(text), want)
    st = plan.stream(max_piece_symbols=30000, record_capacity=1 << 16)
    for off in range(0, text.size, 30000):
        st.feed(text[off:off + 30000])
    assert np.array_equal(st.finish(), want)
    st.close()
    # new keywords: the plan is rebuilt behind the same handle (new symbols get new ids)
    extra = [np.array([text[5], text[6], text[7]], np.uint64), rng.integers(0, 1 << 63, size=2, dtype=np.uint64)]
    for kw in extra:
        m.add_keyword(kw)
        o.add_keyword(kw)
    plan.update(m)
    assert np.array_equal(plan.scan_sorted(dev), o.scan(text))
