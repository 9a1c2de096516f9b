"""Diagnostic: per-wave counters of the 4-gram kernel on the config 3 shape (diag build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
n = 1 << 30
kd, ko = acm.synth.keywords(100000)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((n // 16, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
L = acm.lib(); L.acm_gpu_diag_read.argtypes = [C.c_void_p, C.c_uint]
for co in (True, False):
    for _ in range(2):
        plan.count(text) if co else plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    d = np.zeros((4096, 8), dtype=np.uint64)
    assert L.acm_gpu_diag_read(d.ctypes.data, 4096) == 0
    d = d.astype(np.float64)
    tot = d[:, 0].sum()
    print("count_only=%s: cycles/wave mean %.0f max %.0f | walk %.1f%% (%.0f calls/wave, %.0f cycles/call, %.1f items/call) | consume+issue of first-queue batches %.1f%% incl. walk (%.0f batches/wave, %.0f cycles each)" % (
        co, d[:, 0].mean(), d[:, 0].max(), 100 * d[:, 1].sum() / tot, d[:, 2].mean(), d[:, 1].sum() / max(d[:, 2].sum(), 1),
        d[:, 3].sum() / max(d[:, 2].sum(), 1), 100 * d[:, 5].sum() / tot, d[:, 4].mean(), d[:, 5].sum() / max(d[:, 4].sum(), 1)))
    e = (d[:, 6] - d[:, 6].min()) / 100.0
    eb = e.reshape(256, 16)
    print("   wall clock of wave ends (us after the first): mean %.0f max %.0f; per block (last wave): min %.0f mean %.0f max %.0f; tiles per wave min %d max %d, per block min %d max %d" % (
        e.mean(), e.max(), eb.max(1).min(), eb.max(1).mean(), eb.max(1).max(), d[:, 7].min(), d[:, 7].max(), d[:, 7].reshape(256, 16).sum(1).min(), d[:, 7].reshape(256, 16).sum(1).max()))
    xe = eb.max(1).reshape(32, 8)
    print("   block end by blockIdx % 8:", np.round(xe.mean(0)))
    order = np.argsort(-eb.max(1))[:6]
    print("   slowest blocks:", [(int(b), int(eb.max(1)[b]), int(d[:, 0].reshape(256, 16)[b].max()), int(d[:, 2].reshape(256,16)[b].sum())) for b in order], "(block, end us, max cycles, walk calls)")
