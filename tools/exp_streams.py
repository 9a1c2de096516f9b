"""Experiment driver (not a test): config 2, count-only and record mode, under library builds with
another number of streams per lane (ACM_NATIVE_LIB=.../libac75_amd_s3.so, make exp S=3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aho_corasick_1975_amd as acm

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 1 << 30
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1)
m.add_keywords_packed(kd, ko)
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((1 << 21, 2), dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
plan = m.plan(0)
print(os.environ.get("ACM_NATIVE_LIB", "default"), plan.describe())
for mode in ("count", "record", "count", "record"):
    f = (lambda: plan.count(text, count=cnt)) if mode == "count" else (lambda: plan.scan(text, records=rec, count=cnt))
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    plan.timing(True)
    steps = 40
    t0 = time.perf_counter()
    for _ in range(steps):
        f()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, nl = plan.timing_read()
    plan.timing(False)
    print("  %-6s step %.4f ms  scan kernel %.4f ms  matches %d" % (mode, el / steps * 1e3, ms / nl, int(cnt.item())), flush=True)
plan.status()
