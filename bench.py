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
order) is timed separately and reported as e2e.  `--gpus N` without a launcher (WORLD_SIZE unset)
starts its N workers itself, as fresh child processes, before this process touches a GPU.
Prints ONE JSON line on rank 0.

The default line (config 2, one GPU) also carries, under "other_configs", a short run of configs 3
and 5 at their full sizes after the headline's timed region (a few steps each, same measurement):
the top-level metric / value / config / roofline / cpu_baseline are config 2's and are not touched
by it.  `--no-other-configs` leaves them out.
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
KERNELS = {1: "scan_dense_kernel", 2: "scan_csr_kernel", 3: "scan_sparse_kernel", 4: "scan_starts_kernel", 5: "scan_gram_kernel"}
# what follows the scan kernel in a launch: the kernel that turns parked items / hits into records,
# or the one that closes the holes of a record buffer the scan kernel wrote itself
FOLLOWERS = {1: "expand_items_once_kernel", 4: "expand_hits_kernel", 5: "close_holes_kernel"}
VOCAB = 32768
KNOWN_ANSWERS = os.path.join(ROOT, "tests", "golden", "known_answers.json")


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
    ap.add_argument("--no-other-configs", action="store_true", help="default line only: leave out the short runs of configs 3 and 5")
    ap.add_argument("--cpu-sample-mib", type=int, default=None)
    ap.add_argument("--no-multi-c-abi", action="store_true", help="leave out the e2e leg through acm_gpu_multi_scan_device (C ABI)")
    ap.add_argument("--rccl-c-abi", action="store_true", help="N > 1: also time the e2e job through acm_gpu_comm_gather_records (C ABI over librccl.so); N = 1 runs it anyway")
    ap.add_argument("--no-rccl-c-abi", action="store_true", help="leave that leg out at N = 1 too")
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


def known_answer(config, n_symbols_rank0):
    """(below, count, digest) of the largest committed oracle answer (tests/golden/known_answers.json,
    made by tests/golden/make_known_answers.py in the build container) that lies inside rank 0's text, or None."""
    if not os.path.exists(KNOWN_ANSWERS):
        return None
    with open(KNOWN_ANSWERS) as f:
        ka = json.load(f).get("config%d" % (3 if config == 4 else config))
    if not ka:
        return None
    best = None
    for mk in ka["marks"]:
        if mk["below"] <= n_symbols_rank0 and (best is None or mk["below"] > best[0]):
            best = (int(mk["below"]), int(mk["count"]), int(mk["digest"], 16))
    return best


def measure(ctx, config, cfg, steps, warmup, prewarm_ms, as_configured, want_e2e=True):
    """One workload on this rank's GPU: dictionary, plan, text, timed steps, record-set check.
    Returns (result dict for rank 0's line, objects the CPU baseline needs)."""
    acm, torch, dist = ctx["acm"], ctx["torch"], ctx["dist"]
    rank, world, dev, rehearsal = ctx["rank"], ctx["world"], ctx["dev"], ctx["rehearsal"]
    sym = cfg["sym"]
    # ---- dictionary + plan (host build is not on the metric)
    kd, ko = acm.synth.keywords(cfg["keywords"], sym_bytes=sym, vocab=VOCAB)
    m = acm.Machine(sym)
    t0 = time.time()
    m.add_keywords_packed(kd, ko, ids_as_values=True)
    build_s = time.time() - t0
    plan = m.plan(dev.index)
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
    if prewarm_ms > 0:
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < prewarm_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            prewarm_steps += 8
    for _ in range(warmup):
        step()
    # launches of the scan kernel per step (a 16 GiB step is 8 launches of 2^31 symbols): one untimed step with events
    plan.timing(1)
    step()
    launches_per_step = max(plan.timing_read_all()[2], 1)
    plan.timing(False)
    # The kernel's duration is measured live, by HIP events on the launch stream inside the timed region --
    # around a SAMPLE of its launches: the three events of a launch (start, end of the scan kernel, end of
    # the kernel behind it) cost about 10 us on the stream (tools/exp_timing_overhead.py: a step of config
    # 2 takes 282 us without them, 292 with), which is not the job's time.  About five launches are
    # sampled, an odd period so that they fall on different launches of a step.
    total_launches = launches_per_step * steps
    timing_every = max(1, total_launches // (5 if total_launches >= 20 else 3 if total_launches >= 6 else 2))
    if timing_every > 1 and timing_every % 2 == 0:
        timing_every += 1
    barrier()
    plan.timing(timing_every)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, all_ms, kern_launches = plan.timing_read_all()
    plan.timing(False)
    n_matches = int(count.item())
    assert n_matches == cap - 16, "scan and count-only pass disagree (%d != %d)" % (n_matches, cap - 16)
    plan.status()

    cdev = torch.device("cpu") if rehearsal else dev
    halo_max = halo
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_matches], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_matches = int(tot.item())
        hm = torch.tensor([halo], dtype=torch.int64, device=cdev)
        dist.all_reduce(hm, op=dist.ReduceOp.MAX)      # (rank 0 itself has no halo: the line names the largest any rank used)
        halo_max = int(hm.item())
    else:
        total_matches = n_matches
    total_bytes = float(n_own) * sym * world
    value = total_bytes * steps / elapsed / 1e9

    # whole-set check of this rank's records (device side: count + order-independent digest)
    full_count, full_digest = acm.synth.device_digest(records, n_matches)
    assert full_count == n_matches
    known = known_answer(config, n_own) if (as_configured and rank == 0) else None
    checked = "count-only pass; CPU sample prefix (cpu_baseline)"
    if known:
        below, kc, kdg = known
        got = (full_count, full_digest) if below == n_own else acm.synth.device_digest(records, n_matches, below=below)
        assert got == (kc, kdg), "config %d: records with end_pos < %d differ from the oracle's known answer: %d / %#x, expected %d / %#x" % (
            config, below, got[0], got[1], kc, kdg)
        checked = "oracle (AC-75 variant, tests/golden/make_known_answers.py -> tests/golden/known_answers.json), %s: %d / %#018x" % (
            "whole text" if below == n_own else "records ending in the first %d symbols" % below, kc, kdg)

    # ---- end-to-end leg: scan + canonical order + gather of records to rank 0 (separately timed)
    e2e_obj, gathered = None, None
    if want_e2e:
        e2e_steps = max(1, min(steps, 5)) if n_matches < (1 << 24) else 1

        order_tmp = [None]

        def e2e():     # acm_gpu_scan_ordered_device: scan + order queued together, ONE wait (for the count) at the end
            _, _, order_tmp[0] = plan.scan_ordered(text, n_scan, emit_from=halo, pos_base=pos_base, records=records, count=count, tmp=order_tmp[0])
            n = int(count.item())
            # (the records travel as 8-byte words when their fields fit: Plan.wire / acm_gpu_pack_records_device)
            return acm.sharded.gather_records(records[:n], dst=0, wire=plan.wire(n_scan, pos_base)) if world > 1 else records[:n]

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
        e2e_obj = {
            "what": "scan + canonical order + gather of records to rank 0",
            "value": round(total_bytes * e2e_steps / e2e_elapsed / 1e9, 3), "unit": "GB/s",
            "ms_per_step": round(e2e_elapsed / e2e_steps * 1e3, 4), "steps": e2e_steps,
        }

    # ---- the same end-to-end job through the C ABI's multi-device entry (acm_gpu_multi_scan_device:
    # one process, shard r on device r, ordered records gathered on device 0 by peer copies) -- what a
    # C caller of libac75_amd.so runs.  Rank 0 drives it over all N devices while the other ranks
    # wait at the barrier below; at N = 1 also with 8 shards on the one device.
    c_abi = None
    if want_e2e and ctx.get("multi_c_abi", True) and not rehearsal:
        if rank == 0:
            try:
                c_abi = c_abi_leg(ctx, m, kd, ko, sym, n_own, world, text, total_matches, records if world == 1 else None,
                                  n_matches, 1 if n_matches >= (1 << 24) else 3)
            except Exception as ex:       # the headline line must not depend on this leg
                c_abi = {"error": "%s: %s" % (type(ex).__name__, ex)}
        barrier()
        if e2e_obj is not None and c_abi is not None:
            e2e_obj["c_abi"] = c_abi

    # ---- and through the C ABI's RCCL gather (acm_gpu_comm_*: one process per GPU, the records sent to
    # rank 0 by ncclSend / ncclRecv as 8-byte words) -- what a multi-process C caller runs.  At N = 1 by
    # default (librccl.so itself: communicator, ncclAllGather of the counts); at N > 1 only with
    # --rccl-c-abi: its send / receive path has run over the tests' loopback transport only.
    if want_e2e and not rehearsal and e2e_obj is not None and (world == 1 or ctx.get("rccl_c_abi")) and not ctx.get("no_rccl_c_abi"):
        rccl = None
        try:
            uid = [acm.Comm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            comm = acm.Comm(uid[0], rank, world, 0)
            out = torch.empty((total_matches + 16, 2), dtype=torch.int64, device=cdev) if rank == 0 else None
            r_steps = 1 if n_matches >= (1 << 24) else 3

            def rccl_step():
                _, _, order_tmp[0] = plan.scan_ordered(text, n_scan, emit_from=halo, pos_base=pos_base, records=records, count=count, tmp=order_tmp[0])
                tot, _ = comm.gather_records(plan, records, int(count.item()), pos_base, n_scan, out)
                torch.cuda.synchronize()
                return tot

            assert rccl_step() == total_matches
            barrier()
            t0 = time.perf_counter()
            for _ in range(r_steps):
                rccl_step()
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if rank == 0 and gathered is not None:
                assert acm.synth.device_digest(out, total_matches) == acm.synth.device_digest(gathered, total_matches), "RCCL gather (C ABI): another record set"
            rccl = {"ms_per_step": round(dt / r_steps * 1e3, 4), "value": round(total_bytes * r_steps / dt / 1e9, 3), "unit": "GB/s", "steps": r_steps,
                    "what": "acm_gpu_scan_ordered_device on every rank + acm_gpu_comm_gather_records (C ABI over librccl.so: ncclAllGather of the counts, "
                            "ncclSend / ncclRecv of 8-byte records to rank 0)"}
            del out
            comm.close()
        except Exception as ex:       # the headline line must not depend on this leg
            rccl = {"error": "%s: %s" % (type(ex).__name__, ex)}
        e2e_obj["c_abi_rccl"] = rccl

    res = None
    if rank == 0:
        if gathered is not None:
            assert gathered.shape[0] == total_matches, (gathered.shape[0], total_matches)
            g = gathered.to(dev)
            if g.shape[0] > 1:     # canonical order: end_pos ascending, at one end_pos the longer match first
                ln = g[:, 1] & 0xFFFFFFFF
                ok = (g[1:, 0] > g[:-1, 0]) | ((g[1:, 0] == g[:-1, 0]) & (ln[1:] < ln[:-1]))
                assert bool(ok.all().item()), "gathered records not in canonical order"
                del ln, ok
            if world == 1:         # ... and the same record set as the unordered scan's (whose digest the oracle's known answer pins)
                assert acm.synth.device_digest(g, g.shape[0]) == (full_count, full_digest), "ordered records differ from the scan's record set"
            del g
        # a step is one launch of the scan kernel per segment of 2^31 symbols (16 GiB: 8 launches)
        kern_avg_ms = kern_ms / max(kern_launches, 1)
        follow_avg_ms = (all_ms - kern_ms) / max(kern_launches, 1)
        # sym bytes read per symbol + 16 bytes written per record -- of which the timed
        # scan kernel moves the text and, per record, either the 16-byte record itself
        # (records_direct) or the 8-byte item / hit it parks for the kernel behind it
        direct = bool(info.get("records_direct"))
        rec_bytes_scan = 16.0 if direct else 8.0
        algo_bytes = (float(n_own) * sym + rec_bytes_scan * n_matches) / launches_per_step
        achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
        kname = KERNELS.get(info["kernel"], "kernel %d" % info["kernel"])
        if info["kernel"] == 5 and info.get("variant") == 2:
            kname = "scan_gram2_kernel"     # the 4-gram kernel with the lane-local sieve (dev_gram2.h)
        # HBM bytes per launch of the scan kernel come from separate rocprofv3 --pmc passes of this
        # command (tools/collect_profiles.sh -> profiles/traffic_config<c>.json): counters cannot
        # be read in-process, so this is NOT from the run that prints this line and says so
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_config%d.json" % config)
        if os.path.exists(tpath) and as_configured:
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = "profiles/traffic_config%d.json: %s (separate rocprofv3 --pmc passes of this command, not this run)" % (
                config, tj.get("source", "FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction"))
        if info["kernel"] == 1:
            geometry = "%s<u%d,C=%d,S=%d> %d x %d threads; LDS: %d rows + %d hotfail entries of %d states, %d B" % (
                kname, 8 * info["entry_bytes"], info["chunk_bytes"], info["streams"], info["grid_blocks"],
                info["block_threads"], info["lds_rows"], info["lds_hotfail"], info["dense_rows"], info["lds_bytes"])
        else:
            geometry = "%s %d x %d threads, %d B LDS" % (kname, info["grid_blocks"], info["block_threads"], info["lds_bytes"])
        res = {
            "value": round(value, 3),
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "steps": steps, "warmup": warmup,
            "prewarm": {"ms": prewarm_ms, "untimed_steps": prewarm_steps,
                        "why": "device clock ramp; --prewarm-ms 0 gives the cold-start figure"},
            "dtype": "u%d" % (8 * sym),
            "config": {
                "workload": "BASELINE configs[%d]: %d %s keywords, %d MiB synthetic text per GPU (%d symbols of %d B), "
                            "1 keyword planted per 4096 symbols" % (cfg["idx"], cfg["keywords"], "ASCII" if sym == 1 else "uint32",
                                                                   cfg["mib"], n_own, sym),
                "states": int(m.flatten().info.n_states), "lmax": lmax, "matches_per_gpu": n_matches,
                "matches_total": total_matches,
                "parallelism": "text sharded x%d, %d-symbol halo (the largest any rank used), tables replicated" % (world, halo_max if world > 1 else 0),
                "kernel": geometry, "dictionary_build_s": round(build_s, 3),
                "record_set": {"count": full_count, "digest": "%#018x" % full_digest, "checked_against": checked},
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kname, "kernel_avg_ms": round(kern_avg_ms, 4), "kernel_launches": kern_launches,
                "launches_per_step": launches_per_step,
                "events": "HIP events on the launch stream around every %s launch of the timed region (%d of its %d launches)" % (
                    "" if timing_every == 1 else "%d-th" % timing_every, kern_launches, launches_per_step * steps),
                "algorithmic_bytes_per_launch": algo_bytes,
                "algorithmic_bytes": "%d B per symbol read + %d B per record written by this kernel (%s)" % (
                    sym, int(rec_bytes_scan), "the 16-byte records themselves" if direct else
                    "8-byte items / hits parked for %s, which writes the 16-byte records" % FOLLOWERS.get(info["kernel"], "the kernel behind it")),
                "behind_it": {"kernel": FOLLOWERS.get(info["kernel"]), "avg_ms": round(follow_avg_ms, 4),
                              "what": "HIP events on the launch stream: end of the scan kernel -> end of the kernel it is followed by, averaged over the launches (of a scan's launches only the last one is followed by close_holes_kernel)"},
            },
        }
        if e2e_obj:
            res["e2e"] = e2e_obj
    keep = dict(acm=acm, machine=m, kd=kd, ko=ko, gen=gen, sym=sym, records=records, n_matches=n_matches, n_own=n_own)
    return res, keep


def c_abi_leg(ctx, machine, kd, ko, sym, n_own, world, text0, total_matches, scan_records, n_scan_records, steps):
    """acm_gpu_multi_scan_device over devices 0 .. world - 1 (and, on one GPU, 8 shards on device 0):
    shard r's text is generated on device r as the torch ranks generate theirs; the records arrive on
    device 0 in canonical order and are checked (count, order; on one GPU also the digest of the
    scan's record set).  Returns the timing object for the bench line."""
    acm, torch = ctx["acm"], ctx["torch"]
    n = n_own * world
    out = {}
    for label, devices in ((("devices_0_to_%d" % (world - 1)), list(range(world))),) + (((("eight_shards_on_device_0"), [0] * 8),) if world == 1 else ()):
        mu = acm.binding.MultiScan(machine, devices)
        shards = []
        for r, d in enumerate(devices):
            rb, b, e = mu.shard_bounds(n, r)
            if d == 0 and world == 1:
                shards.append(text0[rb:e])                 # (one GPU: the text is already there)
            elif r == 0:
                shards.append(text0[:e])
            else:
                gen_begin = rb // 4096 * 4096
                with torch.cuda.device(d):
                    g = acm.synth.device_text(e - gen_begin, kd, ko, begin=gen_begin, sym_bytes=sym, vocab=VOCAB, device="cuda:%d" % d)
                shards.append(g[rb - gen_begin:])
        for t in shards:
            assert t.data_ptr() % 16 == 0
        rec = torch.empty((total_matches + 16, 2), dtype=torch.int64, device="cuda:0")
        found = mu.scan_device(shards, n, rec)            # first call: allocates the shards' buffers
        assert found == total_matches, (found, total_matches)
        for d in set(devices):
            torch.cuda.synchronize(d)
        t0 = time.perf_counter()
        for _ in range(steps):
            found = mu.scan_device(shards, n, rec)
        dt = (time.perf_counter() - t0) / steps
        assert found == total_matches
        if found > 1:
            ln = rec[:found, 1] & 0xFFFFFFFF
            ok = (rec[1:found, 0] > rec[:found - 1, 0]) | ((rec[1:found, 0] == rec[:found - 1, 0]) & (ln[1:] < ln[:-1]))
            assert bool(ok.all().item()), "C ABI multi-device scan: records not in canonical order"
            del ln, ok
        if scan_records is not None:
            assert acm.synth.device_digest(rec, found) == acm.synth.device_digest(scan_records, n_scan_records), "C ABI multi-device scan: another record set"
        out[label] = {"ms_per_step": round(dt * 1e3, 4), "value": round(n * sym / dt / 1e9, 3), "unit": "GB/s", "steps": steps,
                      "shards": len(devices)}
        del rec, shards
        mu.close()
        torch.cuda.empty_cache()
    out["what"] = ("acm_gpu_multi_scan_device (C ABI, one process): every shard scanned + ordered on its device, records gathered on "
                   "device 0 by peer copies; host waits inside the call included")
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args))

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
    ctx = dict(acm=acm, torch=torch, dist=dist, rank=rank, world=world, dev=dev, rehearsal=rehearsal, multi_c_abi=not args.no_multi_c_abi,
               rccl_c_abi=args.rccl_c_abi, no_rccl_c_abi=args.no_rccl_c_abi or args.no_multi_c_abi)

    res, keep = measure(ctx, args.config, cfg, args.steps, args.warmup, args.prewarm_ms, as_configured)
    out = None
    if rank == 0:
        out = {
            "metric": "input GB/s scanned, 1k-keyword dictionary, bit-exact match set" if args.config == 2 else
                      "input GB/s scanned, BASELINE config %d, bit-exact match set" % args.config,
            "value": res["value"],
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "prewarm": res["prewarm"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": res["dtype"],
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo; numbers meaningless)" if rehearsal else ""),
            "config": res["config"],
            "roofline": res["roofline"],
            "e2e": res["e2e"],
        }
        if world == 1 and not args.no_cpu_baseline:
            n_sample = min((cfg["cpu_mib"] << 20) // keep["sym"], keep["n_own"])
            out["cpu_baseline"] = cpu_baseline(acm, keep["machine"], keep["kd"], keep["ko"], keep["gen"][:n_sample], keep["sym"],
                                               acm.synth.device_digest(keep["records"], keep["n_matches"], below=n_sample))
        # ---- the other single-GPU configs, briefly, under one extra key (the headline above is final)
        if args.config == 2 and world == 1 and as_configured and not args.no_other_configs:
            del keep
            torch.cuda.empty_cache()
            others = {}
            for c, (k_steps, k_warm) in ((3, (5, 1)), (5, (10, 2))):
                r, k = measure(ctx, c, dict(CONFIGS[c]), k_steps, k_warm, 100, True, want_e2e=True)
                others[str(c)] = {
                    "value": r["value"], "unit": "GB/s", "ms_per_step": r["ms_per_step"], "steps": k_steps, "warmup": k_warm,
                    "workload": r["config"]["workload"], "kernel": r["config"]["kernel"],
                    "matches": r["config"]["matches_per_gpu"], "record_set": r["config"]["record_set"],
                    "roofline": {q: r["roofline"][q] for q in ("frac", "achieved", "kernel", "kernel_avg_ms", "launches_per_step",
                                                               "algorithmic_bytes_per_launch", "behind_it")},
                    "e2e_ms_per_step": r["e2e"]["ms_per_step"],
                }
                del r, k
                torch.cuda.empty_cache()
            out["other_configs"] = others
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
    cores = len(os.sched_getaffinity(0))      # hardware threads this process may run on
    cpu_model, phys = "unknown", set()
    try:
        with open("/proc/cpuinfo") as f:
            pid = None
            for line in f:
                if line.startswith("model name") and cpu_model == "unknown":
                    cpu_model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    pid = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
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
        "cores_are": "hardware threads (os.sched_getaffinity), one scanning thread each", "cpu_model": cpu_model,
        "physical_cores": len(phys) or None,
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
