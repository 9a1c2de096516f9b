"""ctypes bindings of libac75_amd.so (include/acm.h + include/acm_gpu.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libac75_amd.so"

RECORD_DTYPE = np.dtype([("end_pos", "<u8"), ("length", "<u4"), ("keyword_id", "<u4")])

ACM_GPU_OK = 0
ACM_GPU_E_INELIGIBLE = -1
ACM_GPU_E_NODEVICE = -2
ACM_GPU_E_HIP = -3
ACM_GPU_E_OVERFLOW = -4
ACM_GPU_E_ARG = -5


class ACMError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        msg = lib().acm_gpu_strerror(code).decode() if _lib is not None else str(code)
        super().__init__("%s: %s (code %d)" % (what, msg, code))


def library_path():
    # ACM_NATIVE_LIB lets tools/ load the diagnostic build (libac75_amd_diag.so) instead
    return os.environ.get("ACM_NATIVE_LIB") or os.path.join(_HERE, _LIBNAME)


def build_native(force=False):
    """Compile libac75_amd.so in-tree with the committed Makefile (hipcc --offload-arch=gfx950;
    cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-s", "-C", csrc, "clean"], check=True)
    subprocess.run(["make", "-s", "-C", csrc], check=True)
    return library_path()


class MatchHolder(C.Structure):
    _fields_ = [("letters", C.POINTER(C.c_void_p)), ("length", C.c_size_t), ("value", C.c_void_p)]


class FlatInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("sym_bytes", "n_states", "n_keywords", "n_edges", "lmax", "max_outputs",
                                          "alpha_lo", "alpha_span", "width")]


class FlatView(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_uint32)) for n in ("row_ptr", "edge_sym", "edge_next", "fail", "depth", "nb_outputs",
                                                     "term_kw", "out_link", "depth_start", "kw_state")] + [
        ("class_map", C.POINTER(C.c_uint16)), ("edge_letter", C.POINTER(C.c_uint32)), ("class_entries", C.c_uint32),
        ("n_classes", C.c_uint32), ("keys64", C.POINTER(C.c_uint64)), ("n_keys64", C.c_uint32),
        ("keys32", C.POINTER(C.c_uint32)), ("keys32_class", C.POINTER(C.c_uint32)), ("n_keys32", C.c_uint32),
        ("class_rep32", C.POINTER(C.c_uint32))]


class PlanInfo(C.Structure):
    _fields_ = [("device", C.c_int)] + [(n, C.c_uint32) for n in (
        "kernel", "entry_bytes", "width", "dense_rows", "lds_rows", "lds_hotfail", "lds_bytes", "block_threads", "grid_blocks",
        "chunk_bytes", "streams")] + [("table_bytes", C.c_uint64), ("delta_keywords", C.c_uint32), ("merges", C.c_uint32),
                                     ("records_direct", C.c_uint32), ("variant", C.c_uint32)]


_lib = None

# every symbol include/acm.h and include/acm_gpu.h declare
EXPORTS = [
    "ACM_CMP_DEFAULT", "ACM_INCREMENTAL_STRING_MATCHING", "acm_create", "acm_initiate",
    "acm_insert_letter_of_keyword", "acm_insert_end_of_keyword", "acm_match", "acm_matcher_init", "acm_get_match",
    "acm_matcher_release", "acm_nb_keywords", "acm_foreach_keyword", "acm_release", "acm_print",
    "acm_gpu_strerror", "acm_gpu_device_count", "acm_get_keyword", "acm_flatten", "acm_flat_release", "acm_flat_info", "acm_flat_view",
    "acm_flat_dense_rows", "acm_flat_blob_bytes", "acm_flat_to_blob", "acm_flat_from_blob", "acm_flat_save",
    "acm_flat_load", "acm_flat_keyword", "acm_flatten_classes", "acm_gpu_plan_create_classes", "acm_gpu_plan_create", "acm_gpu_plan_create_flat", "acm_gpu_plan_update", "acm_gpu_plan_destroy",
    "acm_gpu_plan_info", "acm_gpu_scan_device", "acm_gpu_count_device", "acm_gpu_sort_tmp_bytes",
    "acm_gpu_sort_records_device", "acm_gpu_order_tmp_bytes", "acm_gpu_order_records_device", "acm_gpu_scan_ordered_tmp_bytes", "acm_gpu_scan_ordered_device", "acm_gpu_scan_host", "acm_scan", "acm_gpu_plan_timing",
    "acm_gpu_plan_timing_read", "acm_gpu_plan_timing_read_all", "acm_gpu_plan_status", "acm_gpu_synth_text",
    "acm_gpu_stream_open", "acm_gpu_stream_feed", "acm_gpu_stream_finish", "acm_gpu_stream_close",
    "acm_gpu_multi_create", "acm_gpu_multi_destroy", "acm_gpu_multi_shard_bounds", "acm_gpu_multi_scan_host",
    "acm_gpu_multi_scan_device", "acm_set_symbol_bytes", "acm_scan_path", "acm_gpu_wire_bits", "acm_gpu_pack_records_device",
    "acm_gpu_unpack_records_device", "acm_gpu_comm_unique_id", "acm_gpu_comm_init_rank", "acm_gpu_comm_free", "acm_gpu_comm_create",
    "acm_gpu_comm_destroy", "acm_gpu_comm_gather_records",
]


def lib():
    """Loads the native library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C aho-corasick-1975_amd/csrc) first" % path)
    L = C.CDLL(path)
    vp, sz, u64, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int
    L.acm_create.restype = vp
    L.acm_create.argtypes = [vp, vp, vp]
    L.acm_initiate.restype = vp
    L.acm_initiate.argtypes = [vp]
    L.acm_insert_letter_of_keyword.restype = None
    L.acm_insert_letter_of_keyword.argtypes = [C.POINTER(vp), vp]
    L.acm_insert_end_of_keyword.restype = vp
    L.acm_insert_end_of_keyword.argtypes = [C.POINTER(vp), vp, vp]
    L.acm_match.restype = sz
    L.acm_match.argtypes = [C.POINTER(vp), vp]
    L.acm_matcher_init.restype = None
    L.acm_matcher_init.argtypes = [C.POINTER(MatchHolder)]
    L.acm_get_match.restype = None
    L.acm_get_match.argtypes = [vp, sz, C.POINTER(MatchHolder)]
    L.acm_matcher_release.restype = None
    L.acm_matcher_release.argtypes = [C.POINTER(MatchHolder)]
    L.acm_nb_keywords.restype = sz
    L.acm_nb_keywords.argtypes = [vp]
    L.acm_foreach_keyword.restype = None
    L.acm_foreach_keyword.argtypes = [vp, vp]
    L.acm_release.restype = None
    L.acm_release.argtypes = [vp]
    L.acm_print.restype = None
    L.acm_print.argtypes = [vp, vp, vp]
    L.acm_gpu_strerror.restype = C.c_char_p
    L.acm_gpu_strerror.argtypes = [i32]
    L.acm_gpu_device_count.restype = i32
    L.acm_get_keyword.restype = i32
    L.acm_get_keyword.argtypes = [vp, u32, C.POINTER(MatchHolder)]
    L.acm_flatten.restype = i32
    L.acm_flatten.argtypes = [vp, C.POINTER(vp)]
    L.acm_flat_release.restype = None
    L.acm_flat_release.argtypes = [vp]
    L.acm_flat_info.restype = None
    L.acm_flat_info.argtypes = [vp, C.POINTER(FlatInfo)]
    L.acm_flat_view.restype = None
    L.acm_flat_view.argtypes = [vp, C.POINTER(FlatView)]
    L.acm_flat_dense_rows.restype = i32
    L.acm_flat_dense_rows.argtypes = [vp, u32, u32, vp]
    L.acm_flatten_classes.restype = i32
    L.acm_flatten_classes.argtypes = [vp, u32, C.POINTER(vp)]
    L.acm_gpu_plan_create_classes.restype = i32
    L.acm_gpu_plan_create_classes.argtypes = [vp, u32, i32, C.POINTER(vp)]
    L.acm_flat_blob_bytes.restype = sz
    L.acm_flat_blob_bytes.argtypes = [vp]
    L.acm_flat_to_blob.restype = i32
    L.acm_flat_to_blob.argtypes = [vp, vp, sz]
    L.acm_flat_from_blob.restype = i32
    L.acm_flat_from_blob.argtypes = [vp, sz, C.POINTER(vp)]
    L.acm_flat_save.restype = i32
    L.acm_flat_save.argtypes = [vp, C.c_char_p]
    L.acm_flat_load.restype = i32
    L.acm_flat_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.acm_flat_keyword.restype = i32
    L.acm_flat_keyword.argtypes = [vp, u32, vp, u32, C.POINTER(u32)]
    L.acm_gpu_plan_create.restype = i32
    L.acm_gpu_plan_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.acm_gpu_plan_update.restype = i32
    L.acm_gpu_plan_update.argtypes = [vp, vp]
    L.acm_gpu_plan_create_flat.restype = i32
    L.acm_gpu_plan_create_flat.argtypes = [vp, i32, C.POINTER(vp)]
    L.acm_gpu_plan_destroy.restype = None
    L.acm_gpu_plan_destroy.argtypes = [vp]
    L.acm_gpu_plan_info.restype = None
    L.acm_gpu_plan_info.argtypes = [vp, C.POINTER(PlanInfo)]
    L.acm_gpu_scan_device.restype = i32
    L.acm_gpu_scan_device.argtypes = [vp, vp, u64, u64, u64, vp, u64, vp, vp]
    L.acm_gpu_count_device.restype = i32
    L.acm_gpu_count_device.argtypes = [vp, vp, u64, u64, vp, vp]
    L.acm_gpu_sort_tmp_bytes.restype = sz
    L.acm_gpu_sort_tmp_bytes.argtypes = [u64]
    L.acm_gpu_sort_records_device.restype = i32
    L.acm_gpu_sort_records_device.argtypes = [vp, vp, u64, vp, sz, vp]
    L.acm_gpu_order_tmp_bytes.restype = sz
    L.acm_gpu_order_tmp_bytes.argtypes = [vp, u64, u64]
    L.acm_gpu_order_records_device.restype = i32
    L.acm_gpu_order_records_device.argtypes = [vp, vp, u64, u64, u64, vp, sz, vp]
    L.acm_gpu_scan_ordered_tmp_bytes.restype = sz
    L.acm_gpu_scan_ordered_tmp_bytes.argtypes = [vp, u64, u64]
    L.acm_gpu_scan_ordered_device.restype = i32
    L.acm_gpu_scan_ordered_device.argtypes = [vp, vp, u64, u64, u64, vp, u64, vp, vp, sz, vp]
    L.acm_gpu_scan_host.restype = i32
    L.acm_gpu_scan_host.argtypes = [vp, vp, u64, u64, u64, vp, u64, C.POINTER(u64)]
    L.acm_scan.restype = i32
    L.acm_scan.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.acm_gpu_wire_bits.restype = i32
    L.acm_gpu_wire_bits.argtypes = [vp, u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.acm_gpu_pack_records_device.restype = i32
    L.acm_gpu_pack_records_device.argtypes = [vp, u64, u64, C.c_uint32, C.c_uint32, vp, vp]
    L.acm_gpu_unpack_records_device.restype = i32
    L.acm_gpu_unpack_records_device.argtypes = [vp, u64, u64, C.c_uint32, C.c_uint32, vp, vp]
    L.acm_set_symbol_bytes.restype = i32
    L.acm_set_symbol_bytes.argtypes = [vp, C.c_uint32]
    L.acm_scan_path.restype = i32
    L.acm_scan_path.argtypes = [vp]
    L.acm_gpu_stream_open.restype = i32
    L.acm_gpu_stream_open.argtypes = [vp, u64, u64, C.POINTER(vp)]
    L.acm_gpu_stream_feed.restype = i32
    L.acm_gpu_stream_feed.argtypes = [vp, vp, u64]
    L.acm_gpu_stream_finish.restype = i32
    L.acm_gpu_stream_finish.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.acm_gpu_stream_close.restype = None
    L.acm_gpu_stream_close.argtypes = [vp]
    L.acm_gpu_plan_status.restype = i32
    L.acm_gpu_plan_status.argtypes = [vp]
    L.acm_gpu_plan_timing.restype = i32
    L.acm_gpu_plan_timing.argtypes = [vp, i32]
    L.acm_gpu_plan_timing_read.restype = i32
    L.acm_gpu_plan_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    L.acm_gpu_multi_create.restype = i32
    L.acm_gpu_multi_create.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.acm_gpu_multi_destroy.restype = None
    L.acm_gpu_multi_destroy.argtypes = [vp]
    L.acm_gpu_multi_shard_bounds.restype = i32
    L.acm_gpu_multi_shard_bounds.argtypes = [vp, u64, C.c_int, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.acm_gpu_multi_scan_host.restype = i32
    L.acm_gpu_multi_scan_host.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.acm_gpu_multi_scan_device.restype = i32
    L.acm_gpu_multi_scan_device.argtypes = [vp, C.POINTER(vp), u64, vp, u64, C.POINTER(u64)]
    L.acm_gpu_comm_unique_id.restype = i32
    L.acm_gpu_comm_unique_id.argtypes = [vp]
    L.acm_gpu_comm_init_rank.restype = i32
    L.acm_gpu_comm_init_rank.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.acm_gpu_comm_free.restype = i32
    L.acm_gpu_comm_free.argtypes = [vp]
    L.acm_gpu_comm_create.restype = i32
    L.acm_gpu_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.acm_gpu_comm_destroy.restype = None
    L.acm_gpu_comm_destroy.argtypes = [vp]
    L.acm_gpu_comm_gather_records.restype = i32
    L.acm_gpu_comm_gather_records.argtypes = [vp, vp, vp, u64, u64, u64, vp, u64, C.POINTER(u64), C.POINTER(u64), vp]
    L.acm_gpu_plan_timing_read_all.restype = i32
    L.acm_gpu_plan_timing_read_all.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(u64)]
    L.acm_gpu_synth_text.restype = i32
    L.acm_gpu_synth_text.argtypes = [i32, vp, u64, u64, u32, u32, vp, vp, u32, vp]
    _lib = L
    return L


def _check(rc, what):
    if rc != ACM_GPU_OK:
        raise ACMError(rc, what)


_SYM_DTYPE = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


class FlatTables:
    """Host snapshot of the machine's flattened goto/failure/output tables (acm_flatten)."""

    def __init__(self, handle):
        self._h = handle
        L = lib()
        info = FlatInfo()
        L.acm_flat_info(handle, C.byref(info))
        self.info = info
        v = FlatView()
        L.acm_flat_view(handle, C.byref(v))
        n, ne = info.n_states, info.n_edges

        def arr(p, cnt):
            return np.ctypeslib.as_array(p, shape=(cnt,)).copy() if cnt else np.zeros(0, np.uint32)
        self.row_ptr = arr(v.row_ptr, n + 1)
        self.edge_sym = arr(v.edge_sym, ne)
        self.edge_next = arr(v.edge_next, ne)
        self.fail = arr(v.fail, n)
        self.depth = arr(v.depth, n)
        self.nb_outputs = arr(v.nb_outputs, n)
        self.term_kw = arr(v.term_kw, n)
        self.out_link = arr(v.out_link, n)
        self.depth_start = arr(v.depth_start, info.lmax + 2)
        self.kw_state = arr(v.kw_state, info.n_keywords)
        # comparator-class machines (acm_flatten_classes): class table and the dictionary's own letters
        self.n_classes = int(v.n_classes)
        self.class_map = np.ctypeslib.as_array(v.class_map, shape=(v.class_entries,)).copy() if v.class_entries else None
        self.edge_letter = arr(v.edge_letter, ne) if v.class_entries else None
        # 8-byte symbols: edge_sym = 1 + index into keys64 (the dictionary's distinct symbols, ascending)
        self.keys64 = np.ctypeslib.as_array(v.keys64, shape=(v.n_keys64,)).copy() if v.n_keys64 else None
        # comparator classes of 4-byte symbols: the dictionary's distinct symbols, their classes (1 .. n_classes),
        # one symbol per class in comparator order
        self.keys32 = arr(v.keys32, v.n_keys32) if v.n_keys32 else None
        self.keys32_class = arr(v.keys32_class, v.n_keys32) if v.n_keys32 else None
        self.class_rep32 = arr(v.class_rep32, v.n_classes) if v.n_keys32 else None
        if v.n_keys32:
            self.edge_letter = arr(v.edge_letter, ne)

    def dense_rows(self, n_rows=None, entry_bytes=None):
        info = self.info
        n_rows = info.n_states if n_rows is None else n_rows
        entry_bytes = (2 if info.n_states <= 32768 else 4) if entry_bytes is None else entry_bytes
        out = np.zeros(n_rows * info.width, dtype=np.uint16 if entry_bytes == 2 else np.uint32)
        _check(lib().acm_flat_dense_rows(self._h, n_rows, entry_bytes, out.ctypes.data), "acm_flat_dense_rows")
        return out.reshape(n_rows, info.width)

    # ---- serialised form (acm_flat_to_blob / acm_flat_from_blob / acm_flat_save / acm_flat_load)
    def to_bytes(self):
        L = lib()
        n = L.acm_flat_blob_bytes(self._h)
        buf = (C.c_ubyte * n)()
        _check(L.acm_flat_to_blob(self._h, buf, n), "acm_flat_to_blob")
        return bytes(buf)

    @classmethod
    def from_bytes(cls, blob):
        h = C.c_void_p()
        buf = (C.c_ubyte * max(len(blob), 1)).from_buffer_copy(blob if len(blob) else b"\0")
        _check(lib().acm_flat_from_blob(buf, len(blob), C.byref(h)), "acm_flat_from_blob")
        return cls(h)

    def save(self, path):
        _check(lib().acm_flat_save(self._h, os.fsencode(path)), "acm_flat_save")

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _check(lib().acm_flat_load(os.fsencode(path), C.byref(h)), "acm_flat_load")
        return cls(h)

    def keyword(self, keyword_id):
        """Spelling of a keyword from the tables alone: numpy array of symbols."""
        sb = self.info.sym_bytes
        n = C.c_uint32(0)
        L = lib()
        _check(L.acm_flat_keyword(self._h, keyword_id, None, 0, C.byref(n)), "acm_flat_keyword")
        out = np.zeros(n.value, dtype={1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[sb])
        _check(L.acm_flat_keyword(self._h, keyword_id, out.ctypes.data, n.value, C.byref(n)), "acm_flat_keyword")
        return out

    def plan(self, device=0):
        """Device plan straight from the tables (no machine needed: acm_gpu_plan_create_flat)."""
        h = C.c_void_p()
        _check(lib().acm_gpu_plan_create_flat(self._h, device, C.byref(h)), "acm_gpu_plan_create_flat")
        return Plan(h, self.info.sym_bytes)

    def close(self):
        if self._h:
            lib().acm_flat_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Machine:
    """An ACMachine over fixed-size symbols compared with ACM_CMP_DEFAULT (reference
    aho_corasick.h:35,45), i.e. what `acm_create (ACM_CMP_DEFAULT, &(size_t){ sym_size }, 0)` returns.
    With `cmp` (a C function pointer of type CMP_TYPE, e.g. ctypes.cast(lib.sym, c_void_p)) the
    machine orders its alphabet with that comparator instead; such a machine reaches the GPU through
    flatten_classes() / plan_classes() when its symbols are 1 or 2 bytes wide."""

    def __init__(self, sym_size=1, cmp=None, cmp_arg=None):
        L = lib()
        self.L = L
        self.sym_size = sym_size
        self.custom_cmp = cmp is not None
        if cmp is None:
            self._arg = C.c_size_t(sym_size)
            cmp = C.c_void_p.in_dll(L, "ACM_CMP_DEFAULT")
            cmp_arg = C.cast(C.pointer(self._arg), C.c_void_p)
        self.handle = L.acm_create(cmp, cmp_arg, None)
        self._keep = []  # letters must outlive the machine (reference aho_corasick.h:39-43)
        self.lmax = 0

    def close(self):
        if self.handle:
            self.L.acm_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _symbols(self, x):
        if isinstance(x, (bytes, bytearray)):
            x = np.frombuffer(bytes(x), dtype=np.uint8)
        return np.ascontiguousarray(x, dtype=_SYM_DTYPE[self.sym_size])

    def add_keyword(self, symbols, value=None):
        """acm_insert_letter_of_keyword per symbol then acm_insert_end_of_keyword.  Returns the
        previous value pointer (None when the keyword had no value yet)."""
        arr = self._symbols(symbols)
        assert arr.size > 0
        self._keep.append(arr)
        L = self.L
        cur = C.c_void_p(L.acm_initiate(self.handle))
        base = arr.ctypes.data
        for i in range(arr.size):
            L.acm_insert_letter_of_keyword(C.byref(cur), base + i * self.sym_size)
        self.lmax = max(self.lmax, int(arr.size))
        return L.acm_insert_end_of_keyword(C.byref(cur), value, None)

    def add_keywords_packed(self, data, offsets, ids_as_values=False):
        """ids_as_values: register value = (void *)(keyword_id + 1) with every keyword, so that a
        caller of the per-symbol API can tell the keywords apart (the reference has no keyword id;
        its MatchHolder.value is the only per-keyword datum, aho_corasick.h:23-28)."""
        data = self._symbols(data)
        self._keep.append(data)
        L = self.L
        base, ss = data.ctypes.data, self.sym_size
        ins, end, init = L.acm_insert_letter_of_keyword, L.acm_insert_end_of_keyword, L.acm_initiate
        nbk = L.acm_nb_keywords
        for k in range(len(offsets) - 1):
            cur = C.c_void_p(init(self.handle))
            ref = C.byref(cur)
            for i in range(int(offsets[k]), int(offsets[k + 1])):
                ins(ref, base + i * ss)
            end(ref, (nbk(self.handle) + 1) if ids_as_values else None, None)
            self.lmax = max(self.lmax, int(offsets[k + 1] - offsets[k]))

    @property
    def nb_keywords(self):
        return int(self.L.acm_nb_keywords(self.handle))

    def match_loop(self, text):
        """The reference's caller loop through the per-symbol host API (examples/test.c:17-23):
        returns records (end_pos, length) and, per record, the matched spelling.  Slow (ctypes
        call per symbol): API-parity tests only."""
        t = self._symbols(text)
        L = self.L
        cur = C.c_void_p(L.acm_initiate(self.handle))
        h = MatchHolder()
        L.acm_matcher_init(C.byref(h))
        out = []
        base = t.ctypes.data
        ptr_t = C.POINTER({1: C.c_uint8, 2: C.c_uint16, 4: C.c_uint32, 8: C.c_uint64}[self.sym_size])
        for i in range(t.size):
            nb = L.acm_match(C.byref(cur), base + i * self.sym_size)
            for j in range(nb):
                L.acm_get_match(cur, j, C.byref(h))
                word = tuple(C.cast(h.letters[k], ptr_t)[0] for k in range(h.length))
                out.append((i, int(h.length), word, h.value))
        L.acm_matcher_release(C.byref(h))
        return out

    def keyword(self, keyword_id):
        """(symbols tuple, value pointer) of a keyword id, through acm_get_keyword."""
        h = MatchHolder()
        self.L.acm_matcher_init(C.byref(h))
        _check(self.L.acm_get_keyword(self.handle, keyword_id, C.byref(h)), "acm_get_keyword")
        ptr_t = C.POINTER({1: C.c_uint8, 2: C.c_uint16, 4: C.c_uint32, 8: C.c_uint64}[self.sym_size])
        word = tuple(C.cast(h.letters[k], ptr_t)[0] for k in range(h.length))
        value = h.value
        self.L.acm_matcher_release(C.byref(h))
        return word, value

    def flatten(self):
        h = C.c_void_p()
        _check(self.L.acm_flatten(self.handle, C.byref(h)), "acm_flatten")
        return FlatTables(h)

    def plan(self, device=0):
        h = C.c_void_p()
        _check(self.L.acm_gpu_plan_create(self.handle, device, C.byref(h)), "acm_gpu_plan_create")
        return Plan(h, self.sym_size)

    # ---- machines with a custom comparator (acm_flatten_classes / acm_gpu_plan_create_classes)
    def flatten_classes(self):
        h = C.c_void_p()
        _check(self.L.acm_flatten_classes(self.handle, self.sym_size, C.byref(h)), "acm_flatten_classes")
        return FlatTables(h)

    def plan_classes(self, device=0):
        h = C.c_void_p()
        _check(self.L.acm_gpu_plan_create_classes(self.handle, self.sym_size, device, C.byref(h)), "acm_gpu_plan_create_classes")
        return Plan(h, self.sym_size)

    def set_symbol_bytes(self, sym_bytes):
        """acm_set_symbol_bytes(): the symbol size of a machine with a comparator of its own."""
        _check(self.L.acm_set_symbol_bytes(self.handle, sym_bytes), "acm_set_symbol_bytes")

    @property
    def scan_path(self):
        """acm_scan_path(): 1 GPU, 2 GPU over comparator classes, 3 the caller loop on the host, 0 none yet."""
        return int(self.L.acm_scan_path(self.handle))

    def scan_host(self, text, capacity=None):
        """acm_scan(): host buffers in, canonical records out (GPU inside; the host loop for machines
        the GPU cannot take by their nature: scan_path says which)."""
        t = np.ascontiguousarray(text) if self.sym_size not in _SYM_DTYPE else self._symbols(text)
        cap = int(capacity) if capacity is not None else max(1024, t.size // 64)
        while True:
            out = np.zeros(cap, dtype=RECORD_DTYPE)
            n = C.c_uint64(0)
            rc = self.L.acm_scan(self.handle, t.ctypes.data, t.size * t.itemsize // self.sym_size, out.ctypes.data, cap, C.byref(n))
            if rc == ACM_GPU_E_OVERFLOW and capacity is None:
                cap = int(n.value)
                continue
            _check(rc, "acm_scan")
            return out[:n.value]


class Plan:
    """Device-resident flattened automaton (ACMPlan).  Scans take torch CUDA tensors (device
    memory and streams are torch's; the kernels are this library's)."""

    def __init__(self, handle, sym_size):
        self.h = handle
        self.sym_size = sym_size
        info = PlanInfo()
        lib().acm_gpu_plan_info(handle, C.byref(info))
        self.info = info

    def close(self):
        if self.h:
            lib().acm_gpu_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, machine):
        """acm_gpu_plan_update(): take over the keywords added to `machine` since this plan was made."""
        _check(lib().acm_gpu_plan_update(self.h, machine.handle), "acm_gpu_plan_update")
        lib().acm_gpu_plan_info(self.h, C.byref(self.info))

    def describe(self):
        i = self.info
        return {n: getattr(i, n) for n, _ in PlanInfo._fields_}

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def scan(self, text, n_symbols=None, emit_from=0, pos_base=0, capacity=None, records=None, count=None):
        """Unsorted scan of a device tensor (uint8 storage of n_symbols * sym_size bytes, or a
        tensor of the symbol dtype).  Returns (records u64[capacity, 2] tensor, count tensor).
        Asynchronous on the current stream."""
        import torch
        assert text.is_cuda and text.is_contiguous()
        if n_symbols is None:
            n_symbols = text.numel() * text.element_size() // self.sym_size
        if records is None:
            cap = int(capacity) if capacity is not None else max(4096, n_symbols // 256)
            records = torch.empty((cap, 2), dtype=torch.int64, device=text.device)
        if count is None:
            count = torch.zeros(1, dtype=torch.int64, device=text.device)
        _check(lib().acm_gpu_scan_device(self.h, text.data_ptr(), n_symbols, emit_from, pos_base, records.data_ptr(),
                                         records.shape[0], count.data_ptr(), self._stream()), "acm_gpu_scan_device")
        return records, count

    def scan_ordered(self, text, n_symbols=None, emit_from=0, pos_base=0, capacity=None, records=None, count=None, tmp=None):
        """acm_gpu_scan_ordered_device(): scan + canonical order, queued on the current stream with
        no host round trip in between (the order passes read the count on the device).  Returns
        (records, count, tmp); records[:count] is in canonical order when count <= capacity."""
        import torch
        assert text.is_cuda and text.is_contiguous()
        if n_symbols is None:
            n_symbols = text.numel() * text.element_size() // self.sym_size
        if records is None:
            cap = int(capacity) if capacity is not None else max(4096, n_symbols // 256)
            records = torch.empty((cap, 2), dtype=torch.int64, device=text.device)
        if count is None:
            count = torch.zeros(1, dtype=torch.int64, device=text.device)
        tb = lib().acm_gpu_scan_ordered_tmp_bytes(self.h, records.shape[0], n_symbols)
        if tmp is None or tmp.numel() < tb:
            tmp = torch.empty(tb, dtype=torch.uint8, device=text.device)
        _check(lib().acm_gpu_scan_ordered_device(self.h, text.data_ptr(), n_symbols, emit_from, pos_base, records.data_ptr(),
                                                 records.shape[0], count.data_ptr(), tmp.data_ptr(), tmp.numel(), self._stream()),
               "acm_gpu_scan_ordered_device")
        return records, count, tmp

    def count(self, text, n_symbols=None, emit_from=0, count=None):
        import torch
        if n_symbols is None:
            n_symbols = text.numel() * text.element_size() // self.sym_size
        if count is None:
            count = torch.zeros(1, dtype=torch.int64, device=text.device)
        _check(lib().acm_gpu_count_device(self.h, text.data_ptr(), n_symbols, emit_from, count.data_ptr(),
                                          self._stream()), "acm_gpu_count_device")
        return count

    def sort(self, records, n, pos_lo=None, span=None):
        """Canonical order (end_pos asc, length desc) of the first n records, in place.  With the
        range of their positions given ([pos_lo, pos_lo + span): what a scan of `span` symbols with
        pos_base = pos_lo leaves) acm_gpu_order_records_device orders them by position buckets and
        LDS sorts; without, acm_gpu_sort_records_device radix-sorts them."""
        import torch
        if n <= 1:
            return records
        if pos_lo is None or span is None:
            tb = lib().acm_gpu_sort_tmp_bytes(n)
            tmp = torch.empty(tb, dtype=torch.uint8, device=records.device)
            _check(lib().acm_gpu_sort_records_device(self.h, records.data_ptr(), n, tmp.data_ptr(), tb, self._stream()),
                   "acm_gpu_sort_records_device")
            return records
        tb = lib().acm_gpu_order_tmp_bytes(self.h, n, span)
        tmp = torch.empty(tb, dtype=torch.uint8, device=records.device)
        _check(lib().acm_gpu_order_records_device(self.h, records.data_ptr(), n, pos_lo, span, tmp.data_ptr(), tb, self._stream()),
               "acm_gpu_order_records_device")
        return records

    def scan_sorted(self, text, n_symbols=None, emit_from=0, pos_base=0, capacity=None, fused=True):
        """Scan + canonical order; grows the record buffer when it overflowed (nothing is dropped).
        fused: acm_gpu_scan_ordered_device (one call), else acm_gpu_scan_device, the count read
        back, acm_gpu_order_records_device.  Returns a numpy structured array (RECORD_DTYPE)."""
        cap = capacity
        while True:
            if fused:
                rec, cnt, _ = self.scan_ordered(text, n_symbols, emit_from, pos_base, cap)
            else:
                rec, cnt = self.scan(text, n_symbols, emit_from, pos_base, cap)
            n = int(cnt.item())
            if n > rec.shape[0]:
                cap = n
                continue
            if not fused:
                ns = n_symbols if n_symbols is not None else text.numel() * text.element_size() // self.sym_size
                self.sort(rec, n, pos_base, ns)
            self.status()
            return np.frombuffer(rec[:n].cpu().numpy().tobytes(), dtype=RECORD_DTYPE).copy()

    def scan_host(self, text, emit_from=0, pos_base=0, capacity=None):
        """acm_gpu_scan_host(): numpy in, numpy out, through the C ABI only (no torch)."""
        t = np.ascontiguousarray(text)
        n_sym = t.size * t.itemsize // self.sym_size
        cap = int(capacity) if capacity is not None else max(1024, n_sym // 64)
        while True:
            out = np.zeros(cap, dtype=RECORD_DTYPE)
            n = C.c_uint64(0)
            rc = lib().acm_gpu_scan_host(self.h, t.ctypes.data, n_sym, emit_from, pos_base, out.ctypes.data, cap,
                                         C.byref(n))
            if rc == ACM_GPU_E_OVERFLOW and capacity is None:
                cap = int(n.value)
                continue
            _check(rc, "acm_gpu_scan_host")
            return out[:n.value]

    def stream(self, max_piece_symbols, record_capacity):
        return Stream(self, max_piece_symbols, record_capacity)

    def wire(self, span, pos_lo):
        """(pos_lo, pos_bits, len_bits) for the 8-byte wire form of the records of a scan of `span` symbols
        with pos_base = pos_lo (acm_gpu_wire_bits), or None when the fields do not fit 64 bits."""
        pb, lb, kb = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        if lib().acm_gpu_wire_bits(self.h, int(span), C.byref(pb), C.byref(lb), C.byref(kb)) != 0:
            return None
        return int(pos_lo), int(pb.value), int(lb.value)

    def status(self):
        """Synchronises and raises if a device-side consistency check failed."""
        _check(lib().acm_gpu_plan_status(self.h), "acm_gpu_plan_status")

    def timing(self, enable=True):
        """True / 1: HIP events around every launch's scan kernel; N > 1: around every N-th launch's; False: off."""
        _check(lib().acm_gpu_plan_timing(self.h, int(enable)), "acm_gpu_plan_timing")

    def timing_read(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(lib().acm_gpu_plan_timing_read(self.h, C.byref(ms), C.byref(n)), "acm_gpu_plan_timing_read")
        return ms.value, int(n.value)

    def timing_read_all(self):
        """(scan kernels' ms, ms from each scan kernel's start to the end of its expansion /
        hole-closing kernel, launches) since timing(True)."""
        ms, allms, n = C.c_double(0), C.c_double(0), C.c_uint64(0)
        _check(lib().acm_gpu_plan_timing_read_all(self.h, C.byref(ms), C.byref(allms), C.byref(n)), "acm_gpu_plan_timing_read_all")
        return ms.value, allms.value, int(n.value)


def pack_records(records, wire):
    """acm_gpu_pack_records_device: int64 [n, 2] CUDA tensor of ordered records -> int64 [n] tensor of 8-byte words."""
    import torch
    pos_lo, pb, lb = wire
    n = records.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=records.device)
    with torch.cuda.device(records.device):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _check(lib().acm_gpu_pack_records_device(records.data_ptr(), n, pos_lo, pb, lb, out.data_ptr(), st), "acm_gpu_pack_records_device")
    return out


def unpack_records(packed, wire, out):
    """acm_gpu_unpack_records_device: int64 [n] CUDA tensor -> the records, into `out` (int64 [n, 2], same device)."""
    import torch
    pos_lo, pb, lb = wire
    assert packed.is_cuda and out.is_cuda and out.shape[0] == packed.shape[0] and out.is_contiguous()
    with torch.cuda.device(out.device):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _check(lib().acm_gpu_unpack_records_device(packed.data_ptr(), packed.shape[0], pos_lo, pb, lb, out.data_ptr(), st), "acm_gpu_unpack_records_device")
    return out


class MultiScan:
    """acm_gpu_multi_*: one process, shard r of a text on devices[r] (a device may repeat), the
    ordered records gathered on devices[0] by peer copies -- the C caller's way to use the GPUs of
    a node (include/acm_gpu.h); torch.distributed jobs use sharded.py instead."""

    def __init__(self, machine, devices):
        self.devices = list(devices)
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _check(lib().acm_gpu_multi_create(machine.handle, arr, len(self.devices), C.byref(h)), "acm_gpu_multi_create")
        self.h = h

    def close(self):
        if self.h:
            lib().acm_gpu_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard_bounds(self, n_symbols, shard):
        rb, b, e = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().acm_gpu_multi_shard_bounds(self.h, n_symbols, shard, C.byref(rb), C.byref(b), C.byref(e)), "acm_gpu_multi_shard_bounds")
        return int(rb.value), int(b.value), int(e.value)

    def scan_host(self, text, capacity=None):
        """numpy text in host memory -> records of the whole text in canonical order (numpy)."""
        t = np.ascontiguousarray(text)
        n = t.size
        cap = int(capacity) if capacity is not None else max(n // 16, 1024)
        while True:
            out = np.zeros(cap, dtype=RECORD_DTYPE)
            found = C.c_uint64(0)
            rc = lib().acm_gpu_multi_scan_host(self.h, t.ctypes.data, n, out.ctypes.data, cap, C.byref(found))
            if rc == -4 and capacity is None:       # ACM_GPU_E_OVERFLOW: the call says how many there are
                cap = int(found.value)
                continue
            _check(rc, "acm_gpu_multi_scan_host")
            return out[:found.value]

    def scan_device(self, shard_tensors, n_symbols, records):
        """shard_tensors[r]: torch tensor on devices[r] holding [read_begin_r, own_end_r); records:
        int64 [cap, 2] tensor on devices[0].  Returns the number of records (all of the text's)."""
        ptrs = (C.c_void_p * len(shard_tensors))(*[int(t.data_ptr()) for t in shard_tensors])
        found = C.c_uint64(0)
        _check(lib().acm_gpu_multi_scan_device(self.h, ptrs, n_symbols, records.data_ptr(), records.shape[0], C.byref(found)),
               "acm_gpu_multi_scan_device")
        return int(found.value)


class Comm:
    """acm_gpu_comm_*: one process (or, in the tests, one thread) per rank; the ranks' ordered
    records gathered on a root over RCCL through the C ABI (include/acm_gpu.h).  `unique_id()` on
    one rank, its 128 bytes handed to the others, then Comm(id, rank, world) on every rank with its
    device current."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().acm_gpu_comm_unique_id(buf), "acm_gpu_comm_unique_id")
        return buf.raw

    def __init__(self, unique_id, rank, world, root=0):
        self.rank, self.world, self.root = int(rank), int(world), int(root)
        self.nccl = C.c_void_p()
        self.h = C.c_void_p()
        _check(lib().acm_gpu_comm_init_rank(C.create_string_buffer(bytes(unique_id), 128), self.rank, self.world, C.byref(self.nccl)),
               "acm_gpu_comm_init_rank")
        _check(lib().acm_gpu_comm_create(self.nccl, self.rank, self.world, self.root, C.byref(self.h)), "acm_gpu_comm_create")

    def gather_records(self, plan, local, n_local, pos_lo, span, out=None, stream=None):
        """local: int64 [*, 2] device tensor with this rank's n_local ordered records (positions in
        [pos_lo, pos_lo + span)); out: int64 [cap, 2] device tensor on the root (None elsewhere).
        Returns (records of all ranks, per-rank counts); raises ACMError(-4) on every rank when the
        root's buffer is too small."""
        total = C.c_uint64(0)
        counts = (C.c_uint64 * self.world)()
        rc = lib().acm_gpu_comm_gather_records(self.h, plan.h if plan is not None else None, local.data_ptr() if n_local else None, int(n_local),
                                               int(pos_lo), int(span), out.data_ptr() if out is not None else None,
                                               out.shape[0] if out is not None else 0, C.byref(total), counts, stream)
        if rc == -4:
            raise ACMError(rc, "acm_gpu_comm_gather_records: %d records, more than the root's buffer holds" % total.value)
        _check(rc, "acm_gpu_comm_gather_records")
        return int(total.value), [int(x) for x in counts]

    def close(self):
        if self.h:
            lib().acm_gpu_comm_destroy(self.h)
            self.h = None
        if self.nccl:
            lib().acm_gpu_comm_free(self.nccl)
            self.nccl = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """acm_gpu_stream_*: feed host buffers piece by piece, get the records of the whole stream."""

    def __init__(self, plan, max_piece_symbols, record_capacity):
        self.plan = plan
        self.capacity = int(record_capacity)
        h = C.c_void_p()
        _check(lib().acm_gpu_stream_open(plan.h, int(max_piece_symbols), self.capacity, C.byref(h)), "acm_gpu_stream_open")
        self.h = h
        self._keep = []

    def feed(self, array):
        """array: contiguous numpy array (or anything exposing ctypes.data and nbytes) of symbols"""
        a = np.ascontiguousarray(array)
        self._keep = (self._keep + [a])[-3:]      # the library reads it asynchronously
        n = a.size * a.itemsize // self.plan.sym_size
        _check(lib().acm_gpu_stream_feed(self.h, a.ctypes.data, n), "acm_gpu_stream_feed")

    def feed_ptr(self, ptr, n_symbols):
        _check(lib().acm_gpu_stream_feed(self.h, ptr, int(n_symbols)), "acm_gpu_stream_feed")

    def finish(self):
        out = np.zeros(self.capacity, dtype=RECORD_DTYPE)
        n = C.c_uint64(0)
        _check(lib().acm_gpu_stream_finish(self.h, out.ctypes.data, self.capacity, C.byref(n)), "acm_gpu_stream_finish")
        return out[:n.value]

    def close(self):
        if self.h:
            lib().acm_gpu_stream_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
