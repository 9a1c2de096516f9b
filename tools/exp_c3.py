"""Experiment driver: config 3 shape (100k byte keywords) scan and count-only times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((n // 16, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for co in (False, True):
    f = (lambda: plan.count(text)) if co else (lambda: plan.scan(text, records=rec, count=cnt))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("K=%d kernel=%d n=%d count_only=%s matches=%d  %.3f ms  %.1f GB/s" % (K, plan.info.kernel, n, co, int(cnt.item()), dt * 1e3, n / dt / 1e9), flush=True)
