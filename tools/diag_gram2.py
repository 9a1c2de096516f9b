"""Diagnostic: per-wave cycle stamps of scan_gram2_kernel on config 3's shape (diag build: make -C csrc diag;
ACM_NATIVE_LIB=.../libac75_amd_diag.so python tools/diag_gram2.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
n = 1 << 31
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
    groups = n / 1024 / 4096
    names = ["total", "top wait", "sieve", "list+rebuild+issue", "consume wait", "consume body", "walk", "steps"]
    print("count_only=%s: per wave and group (cycles): " % co + ", ".join("%s %.0f" % (names[i], d[:, i].mean() / groups) for i in range(7)) +
          ", steps per group %.2f; wait per step %.0f, body per step %.0f" % (d[:, 7].mean() / groups, d[:, 4].sum() / d[:, 7].sum(), d[:, 5].sum() / d[:, 7].sum()), flush=True)
