"""Experiment driver: the e2e step (acm_gpu_scan_ordered_device + the wait for the count) of configs 2 and 5,
a few times each; under `rocprofv3 --kernel-trace` tools/exp_e2e_timeline.py prints the kernels of
the last step with their start offsets."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
which = [int(a) for a in sys.argv[1:]] or [2, 5]
for c in which:
    if c == 2:
        kd, ko = acm.synth.keywords(1000); sb = 1; n = 1 << 30
    else:
        kd, ko = acm.synth.keywords(10000, sym_bytes=4); sb = 4; n = 1 << 28
    m = acm.Machine(sb); m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    text = acm.synth.device_text(n, kd, ko, sym_bytes=sb) if sb > 1 else acm.synth.device_text(n, kd, ko)
    rec = torch.empty((1 << 21, 2), dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    tmp = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _, _, tmp = plan.scan_ordered(text, n, records=rec, count=cnt, tmp=tmp)
            k = int(cnt.item())
        dt = (time.perf_counter() - t0) / 5
        print("config %d: %d records, e2e %.4f ms" % (c, k, dt * 1e3), flush=True)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.scan(text, n, records=rec, count=cnt)
            k = int(cnt.item())
        dt = (time.perf_counter() - t0) / 5
        print("config %d: scan alone + wait %.4f ms" % (c, dt * 1e3), flush=True)
