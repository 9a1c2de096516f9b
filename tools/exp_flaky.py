"""Experiment driver (not a test): repeats one small dense-kernel case whose item regions overflow
(in-place expansion beside parked items) and counts how often the record count deviates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
from tests.cases import small_cases, build_pair

name = sys.argv[1] if len(sys.argv) > 1 else "long_keywords"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kws, text, sym = small_cases()[name]
m, o = build_pair(kws, sym)
want = o.scan(text)
t = np.frombuffer(bytes(text), np.uint8) if isinstance(text, bytes) else text
dev = torch.from_numpy(t.copy()).cuda()
for mode in (os.environ.get("ACM_GPU_EXPAND", "2"),):
    plan = m.plan(0)
    bad = {}
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rec = torch.empty((want.size + 100, 2), dtype=torch.int64, device="cuda")
    for r in range(reps):
        plan.scan(dev, records=rec, count=cnt)
        k = int(cnt.item())
        if k != want.size:
            bad[k - want.size] = bad.get(k - want.size, 0) + 1
        c = int(plan.count(dev).item())
        if c != want.size:
            bad[("count", c - want.size)] = bad.get(("count", c - want.size), 0) + 1
    print(os.environ.get("ACM_NATIVE_LIB", "default")[-24:], "expand", mode, name, "want", want.size, "deviations", bad, flush=True)
