"""PCIe-inclusive rates (never the bench's `value`): streaming scan from pinned host memory and
the one-call host-buffer scan, 1 GiB of config-2 text."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
kd, ko = acm.synth.keywords(1000)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
host = acm.synth.device_text(n, kd, ko).cpu().pin_memory()
for piece_mib in (8, 32, 128):
    st = plan.stream(max_piece_symbols=piece_mib << 20, record_capacity=1 << 21)
    st.feed_ptr(host.data_ptr(), n); got = st.finish()          # warm
    st.close()
    st = plan.stream(max_piece_symbols=piece_mib << 20, record_capacity=1 << 21)
    t0 = time.perf_counter()
    st.feed_ptr(host.data_ptr(), n)
    got = st.finish()
    dt = time.perf_counter() - t0
    st.close()
    print("stream, %3d MiB pieces, pinned host text: %d records, %.1f ms, %.1f GB/s (H2D + scan + sort + D2H of records)" % (
        piece_mib, got.size, dt * 1e3, n / dt / 1e9), flush=True)
arr = host.numpy()
out = np.zeros(1 << 21, dtype=acm.RECORD_DTYPE)
nf = C.c_uint64(0)
for rep in range(2):
    t0 = time.perf_counter()
    rc = acm.lib().acm_scan(m.handle, arr.ctypes.data, n, out.ctypes.data, out.size, C.byref(nf))
    dt = time.perf_counter() - t0
print("acm_scan (one call: hipMalloc + H2D + scan + sort + D2H), rc=%d: %d records, %.1f ms, %.1f GB/s" % (rc, nf.value, dt * 1e3, n / dt / 1e9))
