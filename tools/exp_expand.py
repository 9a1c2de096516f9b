"""Experiment driver (not a test): config 2 step time under the expansion variants
(ACM_GPU_EXPAND: 0 = expand_items_kernel, 1 = waves expand their own queues inside the scan
kernel, 2 = one atomic per block)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 1 << 30
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1)
m.add_keywords_packed(kd, ko)
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((1 << 21, 2), dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "2"]
for rep in range(2):
    for mode in modes:
        os.environ["ACM_GPU_EXPAND"] = mode
        plan = m.plan(0)
        for _ in range(5):
            plan.scan(text, records=rec, count=cnt)
        torch.cuda.synchronize()
        plan.timing(True)
        steps = 50
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.scan(text, records=rec, count=cnt)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms, nl = plan.timing_read()
        k = int(cnt.item())
        r = rec[:k]
        chk = int((r[:, 0] * 1315423911 ^ r[:, 1]).sum().item())
        print("mode %s: step %.4f ms  scan kernel %.4f ms  matches %d chk %x" % (mode, el / steps * 1e3, ms / nl, k, chk & (2**64 - 1)), flush=True)
        plan.status()
        plan.close()
