"""Experiment driver: what a delta plan costs on a BIG text.  acm_gpu_plan_update
keeps the tables of a dense / 4-gram plan and puts the keywords added since into a small delta plan
that is scanned right after the plan over the same buffer: a second pass over the text.  Config 2's
dictionary and 1 GiB text; scan times (records resident, unsorted) with a delta of 0, 1, 64 and 125
keywords, and after the merge the 126th... (here: the 257th keyword past an eighth) triggers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm

K, n = 1000, 1 << 30
kd, ko = acm.synth.keywords(K + 400)
m = acm.Machine(1)
m.add_keywords_packed(kd[:ko[K]], ko[:K + 1])
plan = m.plan(0)
text = acm.synth.device_text(n, kd[:ko[K]], ko[:K + 1])
rec = torch.empty((2_000_000, 2), dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")


def timed(label):
    for _ in range(5):
        plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    d = plan.describe()
    print("%-34s %.3f ms per scan of 1 GiB (%.0f GB/s), %d records, kernel %d, delta keywords %d, merges %d" % (
        label, dt * 1e3, n / dt / 1e9, int(cnt.item()), d["kernel"], d["delta_keywords"], d["merges"]), flush=True)


timed("no delta")
added = 0
for target in (1, 64, 125, 256, 257):
    while added < target:
        k = K + added
        m.add_keyword(kd[ko[k]:ko[k + 1]])
        added += 1
    t0 = time.perf_counter()
    plan.update(m)
    up = (time.perf_counter() - t0) * 1e3
    timed("delta of %d keywords (update %.2f ms)" % (target, up))
