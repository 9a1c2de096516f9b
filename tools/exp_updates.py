"""Experiment driver: cost of a dictionary change between scans (reference generic_test.c:198-229
pattern: insert a keyword, scan, insert, scan ...), split into the host insert
(acm_insert_*: Meyer-85 failure maintenance, not the GPU's business), acm_gpu_plan_update and the
next scan (which writes the pending table patches in front of itself):
  bytes, 1k keywords   -> dense plan, rebuilt behind the same handle
  uint32, 10k keywords -> start-parallel plan, edited in place"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm


def run(name, K, sym, N=300):
    kd, ko = acm.synth.keywords(K + N + 10, sym_bytes=sym)
    m = acm.Machine(sym)
    m.add_keywords_packed(kd[:ko[K]], ko[:K + 1])
    text = acm.synth.device_text(1 << 16, kd, ko, sym_bytes=sym)
    plan = m.plan(0)
    rec = torch.empty((1 << 16, 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")

    def scan():
        plan.scan(text, records=rec, count=cnt)
        torch.cuda.synchronize()
    scan()
    m.add_keyword(kd[ko[K]:ko[K + 1]]); plan.update(m); scan()   # first change: tables move to arrays with headroom
    t = {"host insert": 0.0, "plan update": 0.0, "next scan": 0.0}
    for k in range(K + 1, K + 1 + N):
        a = time.perf_counter(); m.add_keyword(kd[ko[k]:ko[k + 1]]); b = time.perf_counter()
        plan.update(m); c = time.perf_counter(); scan(); d = time.perf_counter()
        t["host insert"] += b - a; t["plan update"] += c - b; t["next scan"] += d - c
    t0 = time.perf_counter()
    for _ in range(N):
        scan()
    base = (time.perf_counter() - t0) / N
    print("%-22s kernel %d: " % (name, plan.info.kernel) + ", ".join("%s %.4f ms" % (k, v / N * 1e3) for k, v in t.items()) +
          "; scan of 64 Ki symbols without a change %.4f ms" % (base * 1e3), flush=True)


run("bytes, 1k keywords", 1000, 1)
run("uint32, 10k keywords", 10000, 4)
if len(sys.argv) > 1 and sys.argv[1] == "big":
    run("bytes, 100k keywords", 100000, 1, N=100)
os.environ["ACM_GPU_DELTA"] = "0"
run("bytes, 1k keywords, rebuild at every update (ACM_GPU_DELTA=0)", 1000, 1, N=100)
