"""ctypes access to oracle/flat_walker.c (TEST INFRASTRUCTURE): scalar CPU walkers over the
flattened tables the GPU consumes."""
import ctypes as C

import numpy as np

from oracle import pyoracle as po


def _lib():
    L = po.lib()
    u64, u32, vp = C.c_uint64, C.c_uint32, C.c_void_p
    L.flatwalk_csr.restype = u64
    L.flatwalk_csr.argtypes = [vp] * 8 + [vp, u64, u32, u64, u64, vp, u64]
    L.flatwalk_dense.restype = u64
    L.flatwalk_dense.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, vp, u64, u64, u64, vp, u64]
    return L


def _arr(text, dtype=None):
    if isinstance(text, (bytes, bytearray)):
        return np.frombuffer(bytes(text), dtype=np.uint8)
    return np.ascontiguousarray(text, dtype=dtype)


def walk_csr(flat, text, emit_from=0, pos_base=0):
    L = _lib()
    t = _arr(text)
    sb = flat.info.sym_bytes
    n = t.size * t.itemsize // sb
    args = [a.ctypes.data for a in (flat.row_ptr, flat.edge_sym, flat.edge_next, flat.fail, flat.nb_outputs,
                                    flat.term_kw, flat.out_link, flat.depth)]
    cnt = L.flatwalk_csr(*args, t.ctypes.data, n, sb, emit_from, pos_base, None, 0)
    out = np.zeros(cnt, dtype=po.RECORD_DTYPE)
    if cnt:
        L.flatwalk_csr(*args, t.ctypes.data, n, sb, emit_from, pos_base, out.ctypes.data, cnt)
    return out


def walk_dense(flat, text, emit_from=0, pos_base=0, n_rows=None, entry_bytes=None):
    L = _lib()
    t = _arr(text, np.uint8)
    rows = flat.dense_rows(n_rows, entry_bytes)
    eb = rows.itemsize
    i = flat.info
    args = [rows.ctypes.data, eb, i.width, i.alpha_lo, i.alpha_span, flat.term_kw.ctypes.data,
            flat.out_link.ctypes.data, flat.depth.ctypes.data, t.ctypes.data, t.size, emit_from, pos_base]
    cnt = L.flatwalk_dense(*args, None, 0)
    out = np.zeros(cnt, dtype=po.RECORD_DTYPE)
    if cnt:
        L.flatwalk_dense(*args, out.ctypes.data, cnt)
    return out
