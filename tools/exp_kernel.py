"""Experiment driver (not a test): times the scan kernel (HIP events around the kernel only)
under input variants, to see where the time goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm


def run(name, K, n, plant=True, steps=5, kw_limit=None):
    kd, ko = acm.synth.keywords(K)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    if plant:
        text = acm.synth.device_text(n, kd, ko)
    else:
        text = acm.synth.device_text(n, kd[:0], ko[:1])
    rec = torch.empty((max(1 << 20, n // 256), 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    plan.timing(True)
    for _ in range(steps):
        plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    ms, nl = plan.timing_read()
    info = plan.describe()
    print("%-34s K=%-6d states=%-7d lds rows=%-5d hotfail=%-5d of %-7d S=%d matches=%-8d kernel %.4f ms  %.1f GB/s" % (
        name, K, m.flatten().info.n_states, info["lds_rows"], info["lds_hotfail"], info["dense_rows"], info["streams"], int(cnt.item()), ms / nl,
        n / (ms / nl * 1e-3) / 1e9), flush=True)


if __name__ == "__main__":
    n = 1 << 30
    run("baseline 1k planted", 1000, n)
    run("1k unplanted text", 1000, n, plant=False)
    run("300 kw (all rows in LDS) planted", 300, n)
    run("300 kw unplanted", 300, n, plant=False)
    run("30 kw unplanted", 30, n, plant=False)
