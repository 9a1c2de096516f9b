"""Experiment driver (not a test): config 2 under the trigram sieve kernel (default) and the dense
kernel (ACM_GPU_SIEVE=0): step time, scan kernel time, count and checksum of the record set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aho_corasick_1975_amd as acm

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1)
m.add_keywords_packed(kd, ko)
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((1 << 22, 2), dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for rep in range(2):
    for sieve in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("1", "0")):
        os.environ["ACM_GPU_SIEVE"] = sieve
        plan = m.plan(0)
        for mode in ("record", "count"):
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
            k = int(cnt.item())
            chk = 0
            if mode == "record":
                r = rec[:k]
                chk = int((r[:, 0] * 1315423911 ^ r[:, 1]).sum().item()) & (2**64 - 1)
            print("K=%d kernel %d %-6s step %.4f ms  scan kernel %.4f ms  %.0f GB/s  matches %d chk %x" % (
                K, plan.info.kernel, mode, el / steps * 1e3, ms / nl, n / (el / steps) / 1e6, k, chk), flush=True)
        plan.status()
        plan.close()
