"""Experiment driver: config 5's shape with 2-byte symbols (10,000 keywords over a 32,768-symbol
vocabulary, uint16 tokens): scan_starts_kernel<uint16_t> count-only and with records.
   python tools/exp_c5_16.py [Mi tokens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aho_corasick_1975_amd as acm
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
kd, ko = acm.synth.keywords(10000, sym_bytes=2, vocab=32768)
m = acm.Machine(2); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
# (acm_gpu_synth_text makes 1- and 4-byte symbols: a 32 Mi-token piece from numpy with keywords planted, tiled on the device)
import numpy as np
rng = np.random.default_rng(3)
piece = rng.integers(0, 32768, size=32 << 20).astype(np.uint16)
for at in rng.integers(0, piece.size - 16, size=piece.size // 4096):
    k = int(rng.integers(0, ko.size - 1))
    w = kd[ko[k]:ko[k + 1]]
    piece[at:at + w.size] = w
text = torch.from_numpy(piece.view(np.int16)).cuda().repeat((n + piece.size - 1) // piece.size)[:n].contiguous()
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
found = int(plan.count(text).item())
rec = torch.empty((found + 4096, 2), dtype=torch.int64, device="cuda")
for co in (False, True):
    f = (lambda: plan.count(text, count=cnt)) if co else (lambda: plan.scan(text, records=rec, count=cnt))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("kernel=%d n=%d tokens of 2 B count_only=%s matches=%d  %.3f ms  %.1f GB/s" % (plan.info.kernel, n, co, int(cnt.item()), dt * 1e3, 2 * n / dt / 1e9), flush=True)
