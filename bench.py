#!/usr/bin/env python3
"""Bench of the hot path: bulk Aho-Corasick scan (BASELINE.json metric: input GB/s scanned,
1k-keyword dictionary, bit-exact match set).

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the scan over one rank's shard, text already resident in HBM, match records
left resident (unsorted) on the owning GPU.  Workloads (BASELINE.json `configs`):

  --config 2 (default, the config the metric is quoted on)  1,000 ASCII keywords, 1 GiB per GPU
  --config 3   100,000 ASCII keywords, 16 GiB per GPU (state table far beyond LDS)
  --config 4   config 3's dictionary, 16 GiB per GPU of ONE 16 x N GiB stream (N = 8: 128 GiB)
  --config 5   uint32 symbols (vocabulary 32,768), 10,000 keywords, 2^30 tokens = 4 GiB per GPU

Weak scaling: every rank owns its share of one stream, scanned with a warm-up halo >= lmax - 1
symbols; no collective inside the timed region.  The gather of records to rank 0 (+ canonical
sort) is timed separately and reported as e2e.  `--gpus N` without a launcher (WORLD_SIZE unset)
starts its N workers itself, as fresh child processes, before this process touches a GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    # keywords, symbol bytes, MiB of text per GPU, MiB (of text) of the CPU sample, BASELINE.json configs[] index
    2: dict(keywords=1000, sym=1, mib=1024, cpu_mib=64, idx=1),
    3: dict(keywords=100000, sym=1, mib=16384, cpu_mib=16, idx=2),
    4: dict(keywords=100000, sym=1, mib=16384, cpu_mib=16, idx=3),
    5: dict(keywords=10000, sym=4, mib=4096, cpu_mib=64, idx=4),
}
KERNELS = {1: "scan_dense_kernel", 2: "scan_csr_kernel", 3: "scan_sparse_kernel", 4: "scan_starts_kernel", 5: "scan_gram_kernel", 6: "scan_sieve_kernel"}
VOCAB = 32768
# known answer of config 2 at full size (1 GiB, rank 0): oracle, AC-75 variant, whole text
CONFIG2_FULL = (555000, 0xdc822ef7f043a221)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--keywords", type=int, default=None, help="override the config's dictionary size")
    ap.add_argument("--mib", type=int, default=None, help="override the config's text MiB per GPU")
    ap.add_argument("--prewarm-ms", type=int, default=300,
                    help="untimed scans before the W warm-up steps until the device has been busy this long (clock ramp), 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=None)
    return ap.parse_args()


def launch_workers(args):
    """--gpus N without a launcher: N fresh child processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment), started before this process has made any GPU
    call; rank 0's JSON line goes to our stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # rank 0's stdout: the JSON line is ours to print; whatever else lands there (the collective
    # backends print connection notes to stdout) goes to stderr
    for line in procs[0].stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def device_digest(torch, rec, n, below=None):
    """(count, digest) of the first n records of an int64 [cap, 2] device buffer, optionally only
    those with end_pos < below.  digest = sum of splitmix64 ((end_pos * 1315423911) ^
    (length << 40) ^ (keyword_id + 1)) mod 2^64 -- order-independent; int64 arithmetic wraps, the
    logical right shifts are spelled with a mask."""
    r = rec[:n]
    pos, lo = r[:, 0], r[:, 1]
    if below is not None:
        keep = pos < below
        pos, lo = pos[keep], lo[keep]
    length = lo & 0xFFFFFFFF
    kw = (lo >> 32) & 0xFFFFFFFF

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def i64(c):  # python int -> the int64 with the same bit pattern
        return c - (1 << 64) if c >= (1 << 63) else c
    x = (pos * 1315423911) ^ (length << 40) ^ (kw + 1)
    x = x + i64(0x9E3779B97F4A7C15)
    x = (x ^ lsr(x, 30)) * i64(0xBF58476D1CE4E5B9)
    x = (x ^ lsr(x, 27)) * i64(0x94D049BB133111EB)
    x = x ^ lsr(x, 31)
    return int(pos.numel()), int(x.sum().item()) & ((1 << 64) - 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    import aho_corasick_1975_amd as acm

    cfg = dict(CONFIGS[args.config])
    if args.keywords is not None:
        cfg["keywords"] = args.keywords
    if args.mib is not None:
        cfg["mib"] = args.mib
    if args.cpu_sample_mib is not None:
        cfg["cpu_mib"] = args.cpu_sample_mib
    sym = cfg["sym"]
    as_configured = args.keywords is None and args.mib is None

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus must equal WORLD_SIZE"
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
    kd, ko = acm.synth.keywords(cfg["keywords"], sym_bytes=sym, vocab=VOCAB)
    m = acm.Machine(sym)
    t0 = time.time()
    m.add_keywords_packed(kd, ko, ids_as_values=True)
    build_s = time.time() - t0
    plan = m.plan(local_rank)
    info = plan.describe()
    lmax = m.lmax

    # ---- this rank's shard of the global stream, generated on the device
    n_own = (cfg["mib"] << 20) // sym                     # symbols per GPU
    own_begin = rank * n_own
    per16 = 16 // sym                                     # the halo keeps the buffer 16-byte aligned
    halo = ((max(lmax - 1, 0) + per16 - 1) // per16) * per16 if own_begin else 0
    gen_begin = (own_begin - halo) // 4096 * 4096 if own_begin else 0    # generator wants 4096-aligned starts
    gen = acm.synth.device_text(n_own + (own_begin - gen_begin), kd, ko, begin=gen_begin, sym_bytes=sym, vocab=VOCAB, device=dev)
    text = gen[own_begin - gen_begin - halo:]
    assert text.data_ptr() % 16 == 0
    n_scan = n_own + halo
    pos_base = own_begin - halo
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    # record capacity from a count-only pass (exact; the scan reports the total needed anyway)
    plan.count(text, n_scan, emit_from=halo, count=count)
    cap = int(count.item()) + 16
    records = torch.empty((cap, 2), dtype=torch.int64, device=dev)

    def step():
        plan.scan(text, n_scan, emit_from=halo, pos_base=pos_base, records=records, count=count)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The device takes tens of milliseconds of load to reach its sustained clock: the first ~50 steps
    # of config 2 (0.3 ms each) run 6-8 % slower than the rest, whatever the kernel.  A scan job of
    # the size the metric is about (config 4: 16 GiB per GPU, ~40 ms) runs at the sustained clock,
    # so the device is kept busy with the same (untimed) step for --prewarm-ms first; reported below.
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            prewarm_steps += 8
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
    assert n_matches == cap - 16, "scan and count-only pass disagree (%d != %d)" % (n_matches, cap - 16)
    plan.status()

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
    total_bytes = float(n_own) * sym * world
    value = total_bytes * args.steps / elapsed / 1e9

    # whole-set check of this rank's records (device side: count + order-independent digest)
    full_count, full_digest = device_digest(torch, records, n_matches)
    assert full_count == n_matches
    known = None
    if args.config == 2 and as_configured and rank == 0:
        known = CONFIG2_FULL
        assert (full_count, full_digest) == known, "config 2 full-size record set differs from the oracle's: %d / %#x" % (
            full_count, full_digest)

    # ---- end-to-end leg: scan + canonical sort + gather of records to rank 0 (separately timed)
    e2e_steps = max(1, min(args.steps, 5)) if n_matches < (1 << 24) else 1

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
        assert gathered.shape[0] == total_matches, (gathered.shape[0], total_matches)
        g = gathered.to(dev)
        if g.shape[0] > 1:
            assert bool((g[1:, 0] >= g[:-1, 0]).all().item()), "gathered records not in canonical order"
        del g
        # a step is one launch of the scan kernel per segment of 2^31 symbols (16 GiB: 8 launches)
        launches_per_step = max(kern_launches // max(args.steps, 1), 1)
        kern_avg_ms = kern_ms / max(kern_launches, 1)
        algo_bytes = (float(n_own) * sym + 16.0 * n_matches) / launches_per_step   # sym B per symbol read + 16 B per record
        achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
        kname = KERNELS.get(info["kernel"], "kernel %d" % info["kernel"])
        # HBM bytes per launch of the scan kernel come from separate rocprofv3 --pmc passes of this
        # command (tools/collect_profiles.sh -> profiles/traffic_config<c>.json): counters cannot
        # be read in-process, so this is NOT from the run that prints this line and says so
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_config%d.json" % args.config)
        if os.path.exists(tpath) and as_configured:
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = "profiles/traffic_config%d.json: %s (separate rocprofv3 --pmc passes of this command, not this run)" % (
                args.config, tj.get("source", "FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction"))
        if info["kernel"] == 1:
            geometry = "%s<u%d,C=%d,S=%d> %d x %d threads; LDS: %d rows + %d hotfail entries of %d states, %d B" % (
                kname, 8 * info["entry_bytes"], info["chunk_bytes"], info["streams"], info["grid_blocks"],
                info["block_threads"], info["lds_rows"], info["lds_hotfail"], info["dense_rows"], info["lds_bytes"])
        else:
            geometry = "%s %d x %d threads, %d B LDS" % (kname, info["grid_blocks"], info["block_threads"], info["lds_bytes"])
        out = {
            "metric": "input GB/s scanned, 1k-keyword dictionary, bit-exact match set" if args.config == 2 else
                      "input GB/s scanned, BASELINE config %d, bit-exact match set" % args.config,
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "prewarm": {"ms": args.prewarm_ms, "untimed_steps": prewarm_steps,
                        "why": "device clock ramp; --prewarm-ms 0 gives the cold-start figure"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u%d" % (8 * sym),
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo; numbers meaningless)" if rehearsal else ""),
            "config": {
                "workload": "BASELINE configs[%d]: %d %s keywords, %d MiB synthetic text per GPU (%d symbols of %d B), "
                            "1 keyword planted per 4096 symbols" % (cfg["idx"], cfg["keywords"], "ASCII" if sym == 1 else "uint32",
                                                                   cfg["mib"], n_own, sym),
                "states": int(m.flatten().info.n_states), "lmax": lmax, "matches_per_gpu": n_matches,
                "matches_total": total_matches,
                "parallelism": "text sharded x%d, %d-symbol halo, tables replicated" % (world, halo if world > 1 else 0),
                "kernel": geometry, "dictionary_build_s": round(build_s, 3),
                "record_set": {"count": full_count, "digest": "%#018x" % full_digest,
                               "checked_against": ("oracle, whole text (AC-75 variant): %d / %#x" % known) if known else
                                                  "count-only pass; CPU sample prefix (cpu_baseline)"},
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kname, "kernel_avg_ms": round(kern_avg_ms, 4), "kernel_launches": kern_launches,
                "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "e2e": {
                "what": "scan + canonical sort + gather of records to rank 0",
                "value": round(total_bytes * e2e_steps / e2e_elapsed / 1e9, 3), "unit": "GB/s",
                "ms_per_step": round(e2e_elapsed / e2e_steps * 1e3, 4), "steps": e2e_steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            n_sample = min((cfg["cpu_mib"] << 20) // sym, n_own)
            out["cpu_baseline"] = cpu_baseline(acm, m, kd, ko, gen[:n_sample], sym,
                                               device_digest(torch, records, n_matches, below=n_sample))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def cpu_baseline(acm, machine, kd, ko, dev_sample, sym, gpu_on_sample):
    """Two CPU legs on a bounded prefix of the rank-0 text, timed on this box's host cores:
      * "port": the oracle (CPU restatement of aho_corasick.c: pointer trie, comparator-ordered edge
        lookup, per-symbol acm_match + acm_get_match loop) -- `value`;
      * "dropin": the product's own host library (libac75_amd.so) driven by the same caller loop
        through tools/libcpuloop.so, one cursor per thread over one shared machine.
    Both also check the GPU records on that prefix (count and digest)."""
    import ctypes as C
    import numpy as np
    from oracle import pyoracle as po
    sample = dev_sample.cpu().numpy()
    if sym == 4:
        sample = sample.view(np.uint32)
    n = sample.size
    nbytes = n * sym
    cores = len(os.sched_getaffinity(0))
    o = po.Oracle(sym, po.MEYER85)
    o.add_keywords_packed(kd, ko)
    t0 = time.perf_counter()
    cnt1, dig1 = o.scan_mt(sample, 1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    cntT, digT = o.scan_mt(sample, cores)
    tT = time.perf_counter() - t0
    ok = (cnt1, dig1) == (cntT, digT) == gpu_on_sample
    assert ok, "GPU records differ from the CPU oracle on the sample prefix: %r %r %r" % ((cnt1, dig1), (cntT, digT), gpu_on_sample)
    res = {
        "value": round(nbytes / tT / 1e9, 5), "unit": "GB/s", "cores": cores, "kind": "port",
        "sample": "first %d MiB of the rank-0 text; oracle/ac_oracle.c (restatement of aho_corasick.c), "
                  "%d threads sharded with lmax-1 overlap; GPU records on the sample verified equal (count + digest)" % (
                      nbytes >> 20, cores),
        "single_thread_value": round(nbytes / t1 / 1e9, 5), "matches_in_sample": int(cnt1),
        "seconds": {"1_thread": round(t1, 2), "%d_threads" % cores: round(tT, 2)},
    }
    lpath = os.path.join(ROOT, "tools", "libcpuloop.so")
    if os.path.exists(lpath):
        C.CDLL(acm.binding.library_path(), mode=C.RTLD_GLOBAL)
        L = C.CDLL(lpath)
        L.acm_cpu_loop.restype = C.c_uint64
        L.acm_cpu_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
        d = C.c_uint64(0)
        t0 = time.perf_counter()
        c1 = L.acm_cpu_loop(machine.handle, sample.ctypes.data, n, sym, machine.lmax, 1, C.byref(d))
        s1 = time.perf_counter() - t0
        d1 = d.value
        t0 = time.perf_counter()
        cT = L.acm_cpu_loop(machine.handle, sample.ctypes.data, n, sym, machine.lmax, cores, C.byref(d))
        sT = time.perf_counter() - t0
        assert (c1, d1) == (cT, d.value) == gpu_on_sample, "the drop-in's own CPU loop differs from the GPU records on the sample"
        res["dropin"] = {
            "what": "libac75_amd.so's own acm_match / acm_get_match loop (tools/cpu_loop.c), same sample, one cursor per "
                    "thread over one shared machine; equals the GPU records (count + digest)",
            "value": round(nbytes / sT / 1e9, 5), "single_thread_value": round(nbytes / s1 / 1e9, 5), "unit": "GB/s", "cores": cores,
            "seconds": {"1_thread": round(s1, 2), "%d_threads" % cores: round(sT, 2)},
        }
    return res


if __name__ == "__main__":
    main()
