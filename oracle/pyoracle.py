"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is a CPU restatement of /root/reference/aho_corasick.c (see ac_oracle.c for the
file:line map).  It checks the product; it is never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

RECORD_DTYPE = np.dtype([("end_pos", "<u8"), ("length", "<u4"), ("keyword_id", "<u4")])

MEYER85 = 0
AC75 = 1

CMP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
DTOR_FN = C.CFUNCTYPE(None, C.c_void_p)


class Holder(C.Structure):
    _fields_ = [("letters", C.POINTER(C.c_void_p)), ("length", C.c_size_t), ("value", C.c_void_p)]


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only, seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("ac_oracle.c", "flat_walker.c", "ac_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        return _LIB
    subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"], check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    L.orc_create.restype = vp
    L.orc_create.argtypes = [vp, vp, vp, C.c_int]
    L.orc_release.argtypes = [vp]
    L.orc_initiate.restype = vp
    L.orc_initiate.argtypes = [vp]
    L.orc_insert_letter_of_keyword.argtypes = [C.POINTER(vp), vp]
    L.orc_insert_end_of_keyword.restype = vp
    L.orc_insert_end_of_keyword.argtypes = [C.POINTER(vp), vp, vp]
    L.orc_match.restype = sz
    L.orc_match.argtypes = [C.POINTER(vp), vp]
    L.orc_matcher_init.argtypes = [C.POINTER(Holder)]
    L.orc_get_match.argtypes = [vp, sz, C.POINTER(Holder)]
    L.orc_matcher_release.argtypes = [C.POINTER(Holder)]
    L.orc_nb_keywords.restype = sz
    L.orc_nb_keywords.argtypes = [vp]
    L.orc_foreach_keyword.argtypes = [vp, vp]
    L.orc_nb_states.restype = sz
    L.orc_nb_states.argtypes = [vp]
    L.orc_state_id.restype = sz
    L.orc_state_id.argtypes = [vp]
    L.orc_state_fail.restype = vp
    L.orc_state_fail.argtypes = [vp]
    L.orc_state_nb_outputs.restype = sz
    L.orc_state_nb_outputs.argtypes = [vp]
    L.orc_state_is_end.restype = C.c_int
    L.orc_state_is_end.argtypes = [vp]
    L.orc_state_depth.restype = sz
    L.orc_state_depth.argtypes = [vp]
    L.orc_scan.restype = u64
    L.orc_scan.argtypes = [vp, vp, u64, sz, u64, u64, vp, u64]
    L.orc_count.restype = u64
    L.orc_count.argtypes = [vp, vp, u64, sz]
    L.orc_digest.restype = u64
    L.orc_digest.argtypes = [vp, u64]
    L.orc_scan_mt.restype = u64
    L.orc_scan_mt.argtypes = [vp, vp, u64, sz, sz, C.c_int, C.POINTER(u64)]
    L.orc_scan_mt_at.restype = u64
    L.orc_scan_mt_at.argtypes = [vp, vp, u64, sz, sz, C.c_int, u64, u64, C.POINTER(u64)]
    L.orc_cmp_default.restype = C.c_int
    _lib = L
    return L


def _sym_dtype(sym_size):
    return {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[sym_size]


class Oracle:
    """One oracle machine over fixed-size symbols compared with the default comparator
    (memcmp over sym_size bytes, reference aho_corasick.c:134-138) or a custom C comparator."""

    def __init__(self, sym_size=1, variant=MEYER85, cmp=None, cmp_arg=None):
        L = lib()
        self.L = L
        self.sym_size = sym_size
        self._keep = []  # letter buffers must outlive the machine (reference aho_corasick.h:39-43)
        if cmp is None:
            self._arg = C.c_size_t(sym_size)
            cmp_ptr = C.cast(L.orc_cmp_default, C.c_void_p)
            arg_ptr = C.cast(C.pointer(self._arg), C.c_void_p)
        else:
            cmp_ptr = C.cast(cmp, C.c_void_p)
            arg_ptr = cmp_arg
        self.m = L.orc_create(cmp_ptr, arg_ptr, None, variant)
        self.lmax = 0
        self.keywords = []  # first-insertion order, distinct

    def close(self):
        if self.m:
            self.L.orc_release(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_keyword(self, symbols):
        """symbols: bytes (sym_size 1) or an integer array.  Registers value = rank + 1 so that the
        scan loop can report keyword ids through the reference API's value pointer.
        Returns (rank, is_new)."""
        arr = np.ascontiguousarray(np.frombuffer(symbols, dtype=np.uint8) if isinstance(symbols, (bytes, bytearray))
                                   else np.asarray(symbols), dtype=_sym_dtype(self.sym_size))
        assert arr.size > 0
        self._keep.append(arr)
        L = self.L
        cur = C.c_void_p(L.orc_initiate(self.m))
        base = arr.ctypes.data
        for i in range(arr.size):
            L.orc_insert_letter_of_keyword(C.byref(cur), base + i * self.sym_size)
        rank = L.orc_nb_keywords(self.m)
        prev = L.orc_insert_end_of_keyword(C.byref(cur), rank + 1, None)
        if prev:
            return int(prev) - 1, False
        self.lmax = max(self.lmax, int(arr.size))
        self.keywords.append(arr)
        return rank, True

    def add_keywords_packed(self, data, offsets):
        """Bulk insert: keyword k = data[offsets[k]:offsets[k+1]] (symbols)."""
        data = np.ascontiguousarray(data, dtype=_sym_dtype(self.sym_size))
        self._keep.append(data)
        L = self.L
        base = data.ctypes.data
        ss = self.sym_size
        for k in range(len(offsets) - 1):
            cur = C.c_void_p(L.orc_initiate(self.m))
            for i in range(int(offsets[k]), int(offsets[k + 1])):
                L.orc_insert_letter_of_keyword(C.byref(cur), base + i * ss)
            rank = L.orc_nb_keywords(self.m)
            prev = L.orc_insert_end_of_keyword(C.byref(cur), rank + 1, None)
            if not prev:
                self.lmax = max(self.lmax, int(offsets[k + 1] - offsets[k]))

    @property
    def nb_keywords(self):
        return int(self.L.orc_nb_keywords(self.m))

    @property
    def nb_states(self):
        return int(self.L.orc_nb_states(self.m))

    def _text(self, text):
        if isinstance(text, (bytes, bytearray)):
            text = np.frombuffer(text, dtype=np.uint8)
        return np.ascontiguousarray(text, dtype=_sym_dtype(self.sym_size))

    def scan(self, text, pos_base=0, emit_from=0):
        t = self._text(text)
        n = self.L.orc_scan(self.m, t.ctypes.data, t.size, self.sym_size, pos_base, emit_from, None, 0)
        out = np.zeros(n, dtype=RECORD_DTYPE)
        if n:
            self.L.orc_scan(self.m, t.ctypes.data, t.size, self.sym_size, pos_base, emit_from, out.ctypes.data, n)
        return out

    def count(self, text):
        t = self._text(text)
        return int(self.L.orc_count(self.m, t.ctypes.data, t.size, self.sym_size))

    def scan_mt(self, text, threads):
        t = self._text(text)
        d = C.c_uint64(0)
        n = self.L.orc_scan_mt(self.m, t.ctypes.data, t.size, self.sym_size, self.lmax, threads, C.byref(d))
        return int(n), int(d.value)

    def scan_mt_at(self, text, threads, pos_base, emit_from):
        """(count, digest) of a piece of a longer text: global positions pos_base + i in the digest,
        matches ending before emit_from (the overlap with the piece before) left out."""
        t = self._text(text)
        d = C.c_uint64(0)
        n = self.L.orc_scan_mt_at(self.m, t.ctypes.data, t.size, self.sym_size, self.lmax, threads, pos_base, emit_from, C.byref(d))
        return int(n), int(d.value)


def digest(records):
    r = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    return int(lib().orc_digest(r.ctypes.data, r.size))
