"""Experiment driver: a big byte dictionary that also has 3-symbol keywords (4-gram kernel with
its short-keyword path against the sticky dense walk, ACM_GPU_GRAM=0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
rng = np.random.default_rng(1)
K = 12000
lens = rng.integers(3, 12, size=K)
data = rng.integers(97, 123, size=int(lens.sum())).astype(np.uint8)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
m = acm.Machine(1); m.add_keywords_packed(data, off)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) << 20
torch.manual_seed(0)
text = torch.randint(97, 123, (n,), dtype=torch.uint8, device="cuda")
plan = m.plan(0)
rec = torch.empty((n // 8, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for co in (False, True):
    f = (lambda: plan.count(text)) if co else (lambda: plan.scan(text, records=rec, count=cnt))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("kernel=%d states=%d n=%d count_only=%s matches=%d  %.3f ms  %.1f GB/s" % (plan.info.kernel, plan.info.dense_rows, n, co, int(cnt.item()), dt * 1e3, n / dt / 1e9), flush=True)
