"""Diagnostic: per-wave counters of the start-parallel kernel on the config 5 shape (diag build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
n = 1 << 28
kd, ko = acm.synth.keywords(10000, sym_bytes=4)
m = acm.Machine(4); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko, sym_bytes=4)
rec = torch.empty((1 << 22, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
L = acm.lib(); L.acm_gpu_diag_read.argtypes = [C.c_void_p, C.c_uint]
for co in (False, True):
    for _ in range(2):
        plan.count(text) if co else plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    d = np.zeros((4096, 8), dtype=np.uint64)
    assert L.acm_gpu_diag_read(d.ctypes.data, 4096) == 0
    d = d.astype(np.float64)
    print("count_only=%s: cycles/wave mean %.0f max %.0f | in walk_starts mean %.0f (%.1f%%) | calls/wave %.1f | cycles/call %.0f | candidates %.0f (%.2f%% of symbols) | deep %.0f (%.4f%%) | tiles/wave %.1f" % (
        co, d[:, 0].mean(), d[:, 0].max(), d[:, 1].mean(), 100 * d[:, 1].sum() / d[:, 0].sum(), d[:, 2].mean(),
        d[:, 1].sum() / max(d[:, 2].sum(), 1), d[:, 3].sum(), 100 * d[:, 3].sum() / n, d[:, 4].sum(), 100 * d[:, 4].sum() / n, d[:, 5].mean()))
