"""Worker of tests/test_gpu_parity.py::test_comm_gather_*_loopback (a process of its own: the
communication library is chosen once per process, ACM_GPU_COMM_LIB).  `world` threads are the ranks
of an acm_gpu_comm_* gather on ONE GPU over tests/helpers/libloopback_comm.so: every rank scans
its shard with its own plan (acm_gpu_scan_ordered_device), the root gathers, and the result must be
the oracle's records of the whole text in canonical order.   usage: worker.py <world> <root> <wire 0|1>"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import aho_corasick_1975_amd as acm
from tests.cases import build_pair

world, root, wire = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1"
if not wire:
    os.environ["ACM_GPU_WIRE"] = "0"
rng = np.random.default_rng(4242 + world)
kws = [rng.integers(97, 105, size=rng.integers(2, 13)).astype(np.uint8) for _ in range(3000)]
text = rng.integers(96, 106, size=(3 << 20) + 77).astype(np.uint8)
m, o = build_pair(kws, 1)
want = o.scan(text)
lmax = max(len(k) for k in kws)
plans = [m.plan(0) for _ in range(world)]
uid = acm.Comm.unique_id()
bounds = [acm.sharded.shard_bounds(text.size, r, world, lmax) for r in range(world)]
results, errors = [None] * world, []


def rank_main(r, capacity):
    try:
        torch.cuda.set_device(0)
        comm = acm.Comm(uid, r, world, root=root)
        rb, b, e = bounds[r]
        dev = torch.from_numpy(text[rb:e]).cuda()
        rec, cnt, _ = plans[r].scan_ordered(dev, emit_from=b - rb, pos_base=rb, capacity=max(want.size, 16))
        torch.cuda.synchronize()
        n = int(cnt.item())
        out = torch.zeros((capacity, 2), dtype=torch.int64, device="cuda") if r == root else None
        try:
            total, counts = comm.gather_records(plans[r], rec, n, rb, e - rb, out)
            torch.cuda.synchronize()
            got = np.frombuffer(out[:total].cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE) if r == root else None
            results[r] = ("ok", total, counts, got, n)
        except acm.ACMError as err:
            results[r] = ("error", err.code, None, None, n)
        comm.close()
    except Exception as ex:     # noqa: BLE001 -- reported by the main thread
        errors.append((r, repr(ex)))


for capacity, expect_overflow in ((want.size + 5, False), (want.size - 1, True)):
    uid = acm.Comm.unique_id()
    threads = [threading.Thread(target=rank_main, args=(r, capacity)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    assert not errors, errors
    if expect_overflow:     # every rank is told, nobody sends
        assert all(res[0] == "error" and res[1] == -4 for res in results), results
        continue
    assert all(res[0] == "ok" for res in results), [res[:2] for res in results]
    total, counts, got, _ = results[root][1:]
    assert total == want.size == sum(counts), (total, want.size, counts)
    assert counts == [res[4] for res in results]
    assert all(res[1] == total and res[2] == counts for res in results)     # every rank knows the totals
    assert np.array_equal(got, want), "gathered records differ from the oracle's"
    assert min(counts) > 100
print("OK world=%d root=%d wire=%s records=%d" % (world, root, wire, want.size))
