"""Experiment driver: config 5 shape scan time (uint32 symbols, 10k keywords)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 18
kd, ko = acm.synth.keywords(10000, sym_bytes=4)
m = acm.Machine(4); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko, sym_bytes=4)
rec = torch.empty((1 << 22, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for co in (False, True):
    f = (lambda: plan.count(text)) if co else (lambda: plan.scan(text, records=rec, count=cnt))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%s kernel=%d n=%d count_only=%s matches=%d  %.3f ms  %.1f GB/s" % (os.environ.get("ACM_NATIVE_LIB", "default")[-16:], plan.info.kernel, n, co, int(cnt.item()), dt * 1e3, n * 4 / dt / 1e9), flush=True)
