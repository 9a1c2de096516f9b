"""Experiment driver: scan_gram2_kernel (lane-local sieve) against scan_gram_kernel on config 3's
dictionary -- the same text, both plans in one process (ACM_GPU_GRAM2 is read when a plan is made):
record sets compared (count + order-independent digest), scan kernel time from the plan's own HIP
events, count-only / records / ordered (tiled).   python tools/exp_gram2.py [MiB] [keywords]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import aho_corasick_1975_amd as acm

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n = mib << 20
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1)
m.add_keywords_packed(kd, ko)
plans = {}
for name, env in (("gram", "0"), ("gram2", "1")):
    os.environ["ACM_GPU_GRAM2"] = env
    m2 = acm.Machine(1)
    m2.add_keywords_packed(kd, ko)
    plans[name] = (m2, m2.plan(0))
    print(name, plans[name][1].describe(), flush=True)
text = acm.synth.device_text(n, kd, ko)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
found = int(plans["gram"][1].count(text).item())
print("text %d MiB, %d records" % (mib, found), flush=True)
rec = torch.empty((found + 4096, 2), dtype=torch.int64, device="cuda")
digests = {}
for name, (_, plan) in plans.items():
    for mode in ("count", "records", "ordered"):
        if mode == "count":
            f = lambda: plan.count(text, count=cnt)
        elif mode == "records":
            f = lambda: plan.scan(text, records=rec, count=cnt)
        else:
            tmp = [None]

            def f():
                _, _, tmp[0] = plan.scan_ordered(text, records=rec, count=cnt, tmp=tmp[0])
        f()
        torch.cuda.synchronize()
        plan.status()
        got = int(cnt.item())
        assert got == found, (name, mode, got, found)
        if mode != "count":
            digests[(name, mode)] = acm.synth.device_digest(rec, got)
        plan.timing(True)
        t0 = time.perf_counter()
        steps = 3
        for _ in range(steps):
            f()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        ms, nl = plan.timing_read()
        plan.timing(False)
        print("%-6s %-8s scan kernel %.4f ms per launch (%d launches)  wall %.3f ms per pass = %.1f GB/s" % (
            name, mode, ms / max(nl, 1), nl, wall * 1e3, n / wall / 1e9), flush=True)
print(digests)
assert len(set(digests.values())) == 1, "record sets differ"
print("record sets equal")
