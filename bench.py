#!/usr/bin/env python3
"""Bench of the hot path: bulk Aho-Corasick scan (BASELINE.json metric: input GB/s scanned,
1k-keyword dictionary, bit-exact match set).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the scan over one rank's shard: BASELINE config 2 (1,000 ASCII keywords of
mean length 8, 1 GiB synthetic text per GPU), text already resident in HBM, match
records left resident (unsorted) on the owning GPU.  Weak scaling: every rank owns 1 GiB of one
N GiB stream, scanned with a 16-byte warm-up halo (>= lmax - 1); no collective inside the timed
region.  The RCCL gather of records to rank 0 (+ canonical sort) is timed separately and
reported as e2e_*.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keywords", type=int, default=1000)
    ap.add_argument("--mib", type=int, default=1024, help="text MiB per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import aho_corasick_1975_amd as acm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU scan path"
    # BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- lets the N>1 code path be
    # exercised on a one-GPU box (numbers from such a run mean nothing and say so)
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- dictionary + plan (host build is not on the metric)
    kd, ko = acm.synth.keywords(args.keywords)
    m = acm.Machine(1)
    t0 = time.time()
    m.add_keywords_packed(kd, ko)
    build_s = time.time() - t0
    plan = m.plan(local_rank)
    lmax = m.lmax

    # ---- this rank's shard of the global stream, generated on the device
    n_own = args.mib << 20
    own_begin = rank * n_own
    halo = 16 if own_begin else 0                       # >= lmax - 1, keeps the buffer 16-byte aligned
    assert lmax - 1 <= 16
    gen_begin = own_begin - 4096 if own_begin else 0    # generator wants 4096-aligned starts
    gen = acm.synth.device_text(n_own + (own_begin - gen_begin), kd, ko, begin=gen_begin, device=dev)
    text = gen[own_begin - gen_begin - halo:]
    n_scan = n_own + halo
    pos_base = own_begin - halo
    cap = max(1 << 20, n_own // 512)
    records = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        plan.scan(text, n_scan, emit_from=halo, pos_base=pos_base, records=records, count=count)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    plan.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_launches = plan.timing_read()
    plan.timing(False)
    n_matches = int(count.item())
    assert n_matches <= cap, "record buffer overflow (%d > %d)" % (n_matches, cap)

    cdev = torch.device("cpu") if rehearsal else dev
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_matches], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_matches = int(tot.item())
    else:
        total_matches = n_matches
    total_bytes = float(n_own) * world
    value = total_bytes * args.steps / elapsed / 1e9

    # ---- end-to-end leg: scan + canonical sort + gather of records to rank 0 (separately timed)
    e2e_steps = max(1, min(args.steps, 5))

    def e2e():
        plan.scan(text, n_scan, emit_from=halo, pos_base=pos_base, records=records, count=count)
        n = int(count.item())
        plan.sort(records, n)
        return acm.sharded.gather_records(records[:n], dst=0) if world > 1 else records[:n]

    gathered = e2e()
    barrier()
    t0 = time.perf_counter()
    for _ in range(e2e_steps):
        gathered = e2e()
    barrier()
    e2e_elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([e2e_elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2e_elapsed = float(t.item())

    out = None
    if rank == 0:
        rec = np.frombuffer(gathered.cpu().numpy().tobytes(), dtype=acm.RECORD_DTYPE)
        assert rec.size == total_matches, (rec.size, total_matches)
        assert np.all(np.diff(rec["end_pos"].astype(np.int64)) >= 0), "gathered records not in canonical order"
        kern_avg_ms = kern_ms / max(kern_launches, 1)
        algo_bytes = float(n_own) + 16.0 * n_matches        # 1 B per symbol read + 16 B per record
        achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
        info = plan.describe()
        # HBM bytes per launch of the scan kernel from the PMC passes of the same command
        # (tools/collect_profiles.sh -> profiles/traffic_latest.json); counters cannot be read in-process
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and args.keywords == 1000 and args.mib == 1024:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "input GB/s scanned, 1k-keyword dictionary, bit-exact match set",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo; numbers meaningless)" if rehearsal else ""),
            "config": {
                "workload": "BASELINE configs[1]: %d ASCII keywords (len 4-12), %d MiB synthetic a-z text per GPU, "
                            "1 keyword planted per 4096 B" % (args.keywords, args.mib),
                "states": int(m.flatten().info.n_states), "lmax": lmax, "matches_per_gpu": n_matches,
                "matches_total": total_matches, "parallelism": "text sharded x%d, 16 B halo, tables replicated" % world,
                "kernel": "scan_dense_kernel<u%d,C=%d,S=%d> %d x %d threads; LDS: %d rows + %d hotfail entries of %d states, %d B" % (
                    8 * info["entry_bytes"], info["chunk_bytes"], info["streams"], info["grid_blocks"],
                    info["block_threads"], info["lds_rows"], info["lds_hotfail"], info["dense_rows"], info["lds_bytes"]),
                "dictionary_build_s": round(build_s, 3),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": "profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction)" if traffic else None,
                "kernel": "scan_dense_kernel", "kernel_avg_ms": round(kern_avg_ms, 4), "kernel_launches": kern_launches,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "e2e": {
                "what": "scan + canonical sort + gather of records to rank 0",
                "value": round(total_bytes * e2e_steps / e2e_elapsed / 1e9, 3), "unit": "GB/s",
                "ms_per_step": round(e2e_elapsed / e2e_steps * 1e3, 4), "steps": e2e_steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(acm, kd, ko, gen, rec, args.cpu_sample_mib)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def cpu_baseline(acm, kd, ko, dev_text, gpu_records, sample_mib):
    """The oracle (CPU restatement of aho_corasick.c: pointer trie, comparator-ordered edge
    lookup, per-symbol acm_match + acm_get_match loop) timed on this box's host cores on a bounded
    prefix of the same text; also cross-checks the GPU records on that prefix."""
    import numpy as np
    from oracle import pyoracle as po
    n = sample_mib << 20
    sample = dev_text[:n].cpu().numpy()
    o = po.Oracle(1, po.MEYER85)
    o.add_keywords_packed(kd, ko)
    t0 = time.perf_counter()
    cnt1, dig1 = o.scan_mt(sample, 1)
    t1 = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    cntT, digT = o.scan_mt(sample, cores)
    tT = time.perf_counter() - t0
    head = gpu_records[gpu_records["end_pos"] < n]
    ok = bool(cnt1 == cntT == head.size and dig1 == digT == po.digest(head))
    assert ok, "GPU records differ from the CPU oracle on the sample prefix"
    return {
        "value": round(n / tT / 1e9, 5), "unit": "GB/s", "cores": cores, "kind": "port",
        "sample": "first %d MiB of the rank-0 text; oracle/ac_oracle.c (restatement of aho_corasick.c), "
                  "%d threads sharded with lmax-1 overlap; GPU records on the sample verified equal" % (sample_mib, cores),
        "single_thread_value": round(n / t1 / 1e9, 5), "matches_in_sample": int(cnt1),
        "seconds": {"1_thread": round(t1, 2), "%d_threads" % cores: round(tT, 2)},
    }


if __name__ == "__main__":
    main()
