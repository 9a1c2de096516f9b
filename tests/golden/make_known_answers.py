#!/usr/bin/env python3
"""Known answers of the BASELINE configs at FULL size, computed on the CPU with the oracle
(oracle/ac_oracle.c, the restatement of /root/reference/aho_corasick.c) -- run in the build
container, results committed as tests/golden/known_answers.json and asserted by the full-size GPU
tests and by bench.py.  Test infrastructure (it drives the oracle): it lives under tests/.

  python tests/golden/make_known_answers.py --config 5            # 2^30 uint32 tokens, 10k keywords
  python tests/golden/make_known_answers.py --config 3 --mib 4096 # first 4 GiB of config 3's 16 GiB

The text is generated piece by piece with the numpy generator (aho-corasick-1975_amd/synth.py, not
the device generator the GPU runs use), every piece scanned with an overlap of lmax - 1 symbols of
its predecessor (warm-up only) by orc_scan_mt_at: count and digest are sums over records, so they
add up over the pieces.  A record belongs to the piece that holds its end position; the running
totals are kept at every GiB of text ("below": records with end_pos < that many symbols).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CONFIGS = {2: dict(keywords=1000, sym=1, mib=1024), 3: dict(keywords=100000, sym=1, mib=16384), 5: dict(keywords=10000, sym=4, mib=4096)}
VOCAB = 32768


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True, choices=sorted(CONFIGS))
    ap.add_argument("--mib", type=int, default=None, help="only the first MiB of the config's text")
    ap.add_argument("--piece-mib", type=int, default=64)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "known_answers.json"))
    args = ap.parse_args()
    import importlib
    synth = importlib.import_module("aho_corasick_1975_amd").synth
    from oracle import pyoracle as po
    cfg = CONFIGS[args.config]
    sym = cfg["sym"]
    mib = args.mib or cfg["mib"]
    n_total = (mib << 20) // sym
    kd, ko = synth.keywords(cfg["keywords"], sym_bytes=sym, vocab=VOCAB)
    o = po.Oracle(sym, po.AC75)
    o.add_keywords_packed(kd, ko)
    overlap = max(o.lmax - 1, 0)
    piece = (args.piece_mib << 20) // sym
    mark_every = (1 << 30) // sym  # a mark per GiB of text
    count, digest = 0, 0
    marks = []
    tail = None
    t0 = time.time()
    for begin in range(0, n_total, piece):
        n = min(piece, n_total - begin)
        body = synth.text(n, kd, ko, begin=begin, sym_bytes=sym, vocab=VOCAB)
        buf = body if tail is None else np.concatenate([tail, body])
        ef = 0 if tail is None else tail.size
        c, d = o.scan_mt_at(buf, args.threads, begin - ef, ef)
        count += c
        digest = (digest + d) & ((1 << 64) - 1)
        tail = body[-overlap:].copy() if overlap else None
        end = begin + n
        if end % mark_every == 0 or end == n_total:
            marks.append({"below": end, "count": count, "digest": "%#018x" % digest})
            print("config %d: %d MiB done, %d records, digest %#018x, %.0f s" % (args.config, end * sym >> 20, count, digest, time.time() - t0), flush=True)
            # written after every mark so that a long run can be cut short and still leave its prefix
            res = {}
            if os.path.exists(args.out):
                with open(args.out) as f:
                    res = json.load(f)
            res["config%d" % args.config] = {
                "keywords": cfg["keywords"], "sym_bytes": sym, "symbols": end, "lmax": o.lmax,
                "oracle": "oracle/ac_oracle.c, AC-75 variant, %d threads, pieces of %d MiB with lmax-1 overlap; text from synth.text (numpy)" % (
                    args.threads, args.piece_mib),
                "complete": end == (cfg["mib"] << 20) // sym, "marks": marks,
            }
            with open(args.out, "w") as f:
                json.dump(res, f, indent=1)
                f.write("\n")


if __name__ == "__main__":
    main()
