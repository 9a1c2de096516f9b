import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
n = int(sys.argv[1]); K = int(sys.argv[2]); reps = int(sys.argv[3])
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko)
rec = torch.empty((max(1 << 20, n // 256), 2), dtype=torch.int64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for r in range(reps):
    plan.scan(text, records=rec, count=cnt)
torch.cuda.synchronize()
try:
    plan.status(); st = "ok"
except Exception as e:
    st = str(e)
print("async x%d n=%d count=%d status=%s" % (reps, n, int(cnt.item()), st), flush=True)
