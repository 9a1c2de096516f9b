"""Experiment: acm_gpu_scan_ordered_device on a text full of matches (a 4-gram plan's crowded tiles), tiled
against ACM_GPU_ORDER=buckets (the three general passes), and the plain scan beside them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
span = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
rng = np.random.default_rng(5)
kws = [rng.integers(97, 97 + span, size=rng.integers(4, 13)).astype(np.uint8) for _ in range(12000)]
m = acm.Machine(1)
for w in kws:
    m.add_keyword(w)
plan = m.plan(0)
text = torch.from_numpy(rng.integers(96, 97 + span + 1, size=n).astype(np.uint8)).cuda()
cnt = plan.count(text); total = int(cnt.item())
print("kernel %d, %d symbols, %d records (%.3f per symbol)" % (plan.info.kernel, n, total, total / n), flush=True)
rec = torch.empty((total + 1024, 2), dtype=torch.int64, device="cuda"); c = torch.zeros(1, dtype=torch.int64, device="cuda")
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("plain scan            %.2f ms" % timeit(lambda: plan.scan(text, records=rec, count=c)), flush=True)
tmp = [None]
def ordered():
    _, _, tmp[0] = plan.scan_ordered(text, records=rec, count=c, tmp=tmp[0])
print("scan_ordered (%s) %.2f ms" % (os.environ.get("ACM_GPU_ORDER", "tiled"), timeit(ordered)), flush=True)
plan.status()
