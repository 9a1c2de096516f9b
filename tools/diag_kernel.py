"""Diagnostic (not a test): per-wave cycle stamps of the dense kernel from the -DACM_DIAG build.
Run with ACM_NATIVE_LIB=aho-corasick-1975_amd/libac75_amd_diag.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_1975_amd as acm


def run(name, K, n, plant=True):
    kd, ko = acm.synth.keywords(K)
    m = acm.Machine(1)
    m.add_keywords_packed(kd, ko)
    plan = m.plan(0)
    text = acm.synth.device_text(n, kd, ko) if plant else acm.synth.device_text(n, kd[:0], ko[:1])
    rec = torch.empty((max(1 << 20, n // 256), 2), dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(2):
        plan.scan(text, records=rec, count=cnt)
    torch.cuda.synchronize()
    waves = 4096
    d = np.zeros((waves, 8), dtype=np.uint64)
    L = acm.lib()
    L.acm_gpu_diag_read.argtypes = [C.c_void_p, C.c_uint]
    assert L.acm_gpu_diag_read(d.ctypes.data, waves) == 0
    d = d.astype(np.float64)
    tot = d[:, 0]
    print("%-28s kernel cycles/wave: mean %.0f max %.0f | items parked/wave %.0f | slow steps %.0f, slow-side cyc %.0f (%.1f%%) = %.0f/step | "
          "text wait %.0f cyc (%.1f%%), %.0f/tile, %d tiles" % (
              name, tot.mean(), tot.max(), d[:, 2].mean(), d[:, 3].mean(), d[:, 4].mean(), 100 * d[:, 4].mean() / tot.mean(),
              d[:, 4].sum() / max(d[:, 3].sum(), 1), d[:, 5].mean(), 100 * d[:, 5].mean() / tot.mean(),
              d[:, 5].sum() / max(d[:, 6].sum(), 1), d[:, 6].mean()), flush=True)


if __name__ == "__main__":
    n = 1 << 30
    run("baseline 1k planted", 1000, n)
    run("1k unplanted", 1000, n, plant=False)
    run("300 kw planted", 300, n)
    run("30 kw unplanted", 30, n, plant=False)
