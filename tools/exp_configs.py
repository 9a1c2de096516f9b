"""Experiment driver: scan time of BASELINE configs 3 (100k keywords, sticky mode) and 5 (uint32) at a given size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm


def run(name, K, n, sym=1, steps=3):
    kd, ko = acm.synth.keywords(K, sym_bytes=sym)
    m = acm.Machine(sym)
    t0 = time.time(); m.add_keywords_packed(kd, ko); tb = time.time() - t0
    t0 = time.time(); plan = m.plan(0); tp = time.time() - t0
    text = acm.synth.device_text(n, kd, ko, sym_bytes=sym)
    rec = torch.empty((max(1 << 20, n // 16), 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    plan.scan(text, records=rec, count=cnt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    plan.status()
    info = plan.describe()
    print("%-28s K=%d sym=%dB n=%d build %.2fs plan %.2fs kernel=%d lds_rows=%d matches=%d  %.3f ms/scan  %.1f GB/s" % (
        name, K, sym, n, tb, tp, info["kernel"], info["lds_rows"], int(cnt.item()), dt * 1e3, n * sym / dt / 1e9), flush=True)


if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    run("config 3 shape (100k kw)", 100000, mib << 20)
    run("config 5 shape (u32 10k kw)", 10000, (mib << 20) // 4, sym=4)
    run("10k kw bytes", 10000, mib << 20)
