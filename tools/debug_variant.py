"""Debug helper: compare a kernel variant with the oracle on the novel and print where records differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm
from oracle import pyoracle as po
text = np.frombuffer(open("tests/golden/mrs_dalloway.txt", "rb").read(), np.uint8)
kws = [b"he", b"she", b"his", b"hers"]
m = acm.Machine(1); o = po.Oracle(1)
for k in kws: m.add_keyword(k); o.add_keyword(k)
plan = m.plan(0)
print(plan.describe())
want = o.scan(text)
dev = torch.from_numpy(text.copy()).cuda()
rec, cnt = plan.scan(dev, capacity=1 << 16)
torch.cuda.synchronize()
n = int(cnt.item())
print("count", n, "want", want.size)
try:
    plan.status()
except Exception as e:
    print("STATUS", e)
got = np.frombuffer(rec[:min(n, rec.shape[0])].cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE).copy()
gs = set(map(tuple, got.tolist())); ws = set(map(tuple, want.tolist()))
miss = sorted(ws - gs); extra = sorted(gs - ws)
print("missing", len(miss), miss[:20])
print("extra", len(extra), extra[:20])
if miss:
    mp = np.array([x[0] for x in miss])
    print("missing pos mod 32:", np.bincount(mp % 32, minlength=32).tolist())
    print("missing pos mod 8192 //2048:", np.bincount((mp % 8192) // 2048, minlength=4).tolist())

# second case: synthetic 1k dictionary, growing sizes, checked through the guard
kd, ko = acm.synth.keywords(1000)
m2 = acm.Machine(1); m2.add_keywords_packed(kd, ko)
o2 = po.Oracle(1); o2.add_keywords_packed(kd, ko)
p2 = m2.plan(0)
for n in (1 << 16, 1 << 20, 1 << 22):
    t = acm.synth.text(n, kd, ko)
    rec, cnt = p2.scan(torch.from_numpy(t).cuda(), capacity=1 << 18)
    torch.cuda.synchronize()
    c = int(cnt.item())
    w = o2.scan(t)
    g = np.frombuffer(rec[:min(c, rec.shape[0])].cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE)
    gs = set(map(tuple, g.tolist())); ws = set(map(tuple, w.tolist()))
    try:
        p2.status(); st = "ok"
    except Exception as e:
        st = str(e)
    print("synthetic n=%d count %d want %d missing %d extra %d status %s" % (n, c, w.size, len(ws - gs), len(gs - ws), st))
    ex = sorted(gs - ws)[:8]; mi = sorted(ws - gs)[:8]
    print("  extra", ex, "missing", mi)
