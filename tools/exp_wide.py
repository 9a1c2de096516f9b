"""Experiment driver: a big byte dictionary over a wide alphabet (64 symbols, 50k keywords of 4-12
symbols): 4-gram kernel with hashed windows against the sticky dense walk (ACM_GPU_GRAM=0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
rng = np.random.default_rng(1)
K, A = 50000, 64
lens = rng.integers(4, 13, size=K)
data = rng.integers(48, 48 + A, size=int(lens.sum())).astype(np.uint8)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
m = acm.Machine(1); m.add_keywords_packed(data, off)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
torch.manual_seed(0)
text = torch.randint(48, 48 + A, (n,), dtype=torch.uint8, device="cuda")
# plant a keyword every 4 KiB
kw_at = torch.arange(0, n - 4096, 4096, device="cuda")
plan = m.plan(0)
rec = torch.empty((n // 64, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for co in (False, True):
    f = (lambda: plan.count(text)) if co else (lambda: plan.scan(text, records=rec, count=cnt))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("kernel=%d states=%d n=%d count_only=%s matches=%d  %.3f ms  %.1f GB/s" % (plan.info.kernel, plan.info.dense_rows, n, co, int(cnt.item()), dt * 1e3, n / dt / 1e9), flush=True)
