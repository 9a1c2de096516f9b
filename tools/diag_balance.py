"""Diagnostic: spread of per-wave kernel cycles within and across workgroups (diag build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
kd, ko = acm.synth.keywords(1000)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
n = 1 << 30
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((1 << 22, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(3):
    plan.scan(text, records=rec, count=cnt)
torch.cuda.synchronize()
d = np.zeros((4096, 8), dtype=np.uint64)
L = acm.lib(); L.acm_gpu_diag_read.argtypes = [C.c_void_p, C.c_uint]
assert L.acm_gpu_diag_read(d.ctypes.data, 4096) == 0
t = d[:, 0].astype(np.float64).reshape(256, 16)
print("per-wave cycles: mean %.0f  min %.0f  max %.0f" % (t.mean(), t.min(), t.max()))
print("per-block: mean of block max %.0f, max of block max %.0f, mean of (block max / block mean) %.3f" % (
    t.max(1).mean(), t.max(1).max(), (t.max(1) / t.mean(1)).mean()))
print("block means: min %.0f max %.0f" % (t.mean(1).min(), t.mean(1).max()))
w0 = d[:, 7].astype(np.float64); w1 = d[:, 1].astype(np.float64)
base = w0.min()
e = (w1 - base).reshape(256, 16) / 100.0   # us at 100 MHz
print("wall clock (us): starts spread %.1f; wave ends: min %.1f mean %.1f max %.1f" % ((w0.max() - base) / 100.0, e.min(), e.mean(), e.max()))
print("block end (max over waves): min %.1f mean %.1f max %.1f" % (e.max(1).min(), e.max(1).mean(), e.max(1).max()))
print("tiles per wave: min %d max %d; per block: min %d max %d" % (d[:, 6].min(), d[:, 6].max(), d[:, 6].reshape(256, 16).sum(1).min(), d[:, 6].reshape(256, 16).sum(1).max()))
xe = e.max(1).reshape(32, 8)
print("end by blockIdx%8 (XCD):", np.round(xe.mean(0), 1))
