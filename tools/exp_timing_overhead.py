"""A step of config 2 (1 GiB, 1,000 keywords) with and without the HIP events acm_gpu_plan_timing records around
every launch: what the measurement costs the thing measured.   python tools/exp_timing_overhead.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import aho_corasick_1975_amd as acm
n = 1 << 30
kd, ko = acm.synth.keywords(1000)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
found = int(plan.count(text).item())
rec = torch.empty((found + 16, 2), dtype=torch.int64, device="cuda")
def step(): plan.scan(text, records=rec, count=cnt)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(8): step()
    torch.cuda.synchronize()
for rep in range(3):
    for on in (False, True):
        plan.timing(on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        if on: plan.timing_read_all()
        plan.timing(False)
        print("timing=%s  %.4f ms per step  %.1f GB/s" % (on, dt * 1e3, n / dt / 1e9), flush=True)
