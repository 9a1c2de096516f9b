"""Synthetic workloads of SURVEY.md section 8(d): counter-based, so the CPU (numpy), the oracle
and the GPU (acm_gpu_synth_text) regenerate identical keywords and text from an index alone.

  sm(x) = splitmix64 finaliser
  byte keywords : len = 4 + sm(1000003 k + 1) % 9, kw[k][j] = 'a' + sm(64 k + j + 7777) % 26
  byte text     : text[i] = 'a' + sm(i + 42) % 26, then one keyword planted per 4096-block:
                  keyword sm(p + 99) % K at offset p + sm(p) % 4080
  uint32 (cfg 5): vocab V, len = 2 + sm(1000003 k + 1) % 7, kw[k][j] = sm(64 k + j + 7777) % V,
                  tok[i] = sm(i + 42) % V, same planting rule
"""
import numpy as np

P = 4096
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def sm(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def keywords(K, sym_bytes=1, vocab=32768):
    """Returns (data, offsets): keyword k = data[offsets[k]:offsets[k+1]]."""
    k = np.arange(K, dtype=np.uint64)
    with np.errstate(over="ignore"):
        if sym_bytes == 1:
            lens = (4 + sm(np.uint64(1000003) * k + np.uint64(1)) % np.uint64(9)).astype(np.int64)
        else:
            lens = (2 + sm(np.uint64(1000003) * k + np.uint64(1)) % np.uint64(7)).astype(np.int64)
        off = np.zeros(K + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        kk = np.repeat(k, lens)
        jj = np.arange(off[-1], dtype=np.uint64) - np.repeat(off[:-1].astype(np.uint64), lens)
        h = sm(np.uint64(64) * kk + jj + np.uint64(7777))
    if sym_bytes == 1:
        data = (ord("a") + h % np.uint64(26)).astype(np.uint8)
    else:
        data = (h % np.uint64(vocab)).astype(np.uint32)
    return data, off.astype(np.uint32)


def text(n, kw_data, kw_off, begin=0, sym_bytes=1, vocab=32768):
    """text[begin : begin + n) of the global stream; begin must be a multiple of 4096."""
    assert begin % P == 0
    i = np.arange(begin, begin + n, dtype=np.uint64)
    if sym_bytes == 1:
        out = (ord("a") + sm(i + np.uint64(42)) % np.uint64(26)).astype(np.uint8)
    else:
        out = (sm(i + np.uint64(42)) % np.uint64(vocab)).astype(np.uint32)
    K = len(kw_off) - 1
    if K:
        p = np.arange(begin, begin + n, P, dtype=np.uint64)
        off = p + sm(p) % np.uint64(P - 16)
        kid = (sm(p + np.uint64(99)) % np.uint64(K)).astype(np.int64)
        for b in range(p.size):
            a, e = int(kw_off[kid[b]]), int(kw_off[kid[b] + 1])
            lo = int(off[b]) - begin
            hi = min(lo + (e - a), n)
            if lo < n:
                out[lo:hi] = kw_data[a:a + (hi - lo)]
    return out


def device_text(n, kw_data, kw_off, begin=0, sym_bytes=1, vocab=32768, device=None):
    """The same stream generated on the GPU by acm_gpu_synth_text; returns a torch tensor."""
    import ctypes as C
    import torch
    from .binding import lib, _check
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    dt = torch.uint8 if sym_bytes == 1 else torch.int32
    out = torch.empty(n, dtype=dt, device=device)
    kd = torch.from_numpy(np.ascontiguousarray(kw_data).view(np.uint8 if sym_bytes == 1 else np.int32)).to(device)
    ko = torch.from_numpy(np.ascontiguousarray(kw_off, dtype=np.uint32).view(np.int32)).to(device)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _check(lib().acm_gpu_synth_text(device.index or 0, out.data_ptr(), n, begin, sym_bytes, vocab, kd.data_ptr(),
                                    ko.data_ptr(), len(kw_off) - 1, st), "acm_gpu_synth_text")
    return out


def device_digest(rec, n, below=None, at_least=None):
    """(count, digest) of the first n records of an int64 [cap, 2] device buffer, optionally only
    those with at_least <= end_pos < below.  digest = sum of splitmix64 ((end_pos * 1315423911) ^
    (length << 40) ^ (keyword_id + 1)) mod 2^64 (oracle/ac_oracle.c record_hash)
    -- order-independent; int64 arithmetic wraps, the logical right shifts are spelled with a mask."""
    r = rec[:n]
    pos, lo = r[:, 0], r[:, 1]
    if below is not None or at_least is not None:
        keep = (pos < below) if below is not None else (pos >= at_least)
        if below is not None and at_least is not None:
            keep = keep & (pos >= at_least)
        pos, lo = pos[keep], lo[keep]
    length = lo & 0xFFFFFFFF
    kw = (lo >> 32) & 0xFFFFFFFF

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def i64(c):  # python int -> the int64 with the same bit pattern
        return c - (1 << 64) if c >= (1 << 63) else c
    x = (pos * 1315423911) ^ (length << 40) ^ (kw + 1)
    x = x + i64(0x9E3779B97F4A7C15)
    x = (x ^ lsr(x, 30)) * i64(0xBF58476D1CE4E5B9)
    x = (x ^ lsr(x, 27)) * i64(0x94D049BB133111EB)
    x = x ^ lsr(x, 31)
    return int(pos.numel()), int(x.sum().item()) & ((1 << 64) - 1)
