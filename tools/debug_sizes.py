import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
from oracle import pyoracle as po
n = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
kd, ko = acm.synth.keywords(K)
m = acm.Machine(1); m.add_keywords_packed(kd, ko)
plan = m.plan(0)
text = acm.synth.device_text(n, kd, ko)
rec, cnt = plan.scan(text, capacity=max(1 << 20, n // 256))
torch.cuda.synchronize()
c = int(cnt.item())
try:
    plan.status(); st = "ok"
except Exception as e:
    st = str(e)
print("n=%d K=%d count=%d status=%s" % (n, K, c, st), flush=True)
