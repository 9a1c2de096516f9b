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


def novel():
    """generic_test.c:166-239 on bytes: the dictionary built up from the novel while it is scanned
    (6,966 inserts), every insert followed by acm_gpu_plan_update and a scan of 3 KB around it"""
    import re
    raw = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mrs_dalloway.txt"), "rb").read()
    t = np.frombuffer(raw, np.uint8).copy()
    up = (t >= 65) & (t <= 90)
    t[up] += 32
    t[~((t >= 97) & (t <= 122))] = 32
    m = acm.Machine(1)
    m.add_keyword(b" the ")
    plan = m.plan(0)
    seen, tu, ts, th, n, kernels = {b"the"}, 0.0, 0.0, 0.0, 0, {}
    for mt in re.finditer(rb"[a-z]+", t.tobytes()):
        w = mt.group(0)
        if w in seen:
            continue
        seen.add(w)
        a = time.perf_counter(); m.add_keyword(b" " + w + b" "); b = time.perf_counter()
        plan.update(m); c = time.perf_counter()
        plan.scan_host(t[max(mt.start() - 1500, 0):mt.end() + 1500]); d = time.perf_counter()
        th += b - a; tu += c - b; ts += d - c; n += 1
        kernels[plan.info.kernel] = kernels.get(plan.info.kernel, 0) + 1
    plan.info_refresh() if hasattr(plan, "info_refresh") else None
    print("novel replay: %d inserts; per insert: host %.1f us, plan update %.1f us, scan_host of 3 KB %.1f us; merges %d; kernel of the main plan at each insert %s" % (
        n, th / n * 1e6, tu / n * 1e6, ts / n * 1e6, plan.describe()["merges"], kernels), flush=True)


novel()
run("bytes, 1k keywords", 1000, 1)
run("uint32, 10k keywords", 10000, 4)
if len(sys.argv) > 1 and sys.argv[1] == "big":
    run("bytes, 100k keywords", 100000, 1, N=100)
os.environ["ACM_GPU_DELTA"] = "0"
run("bytes, 1k keywords, rebuild at every update (ACM_GPU_DELTA=0)", 1000, 1, N=100)
