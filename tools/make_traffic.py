#!/usr/bin/env python3
"""profiles/traffic_config<c>.json (what bench.py's roofline.traffic reads) from a summary written by
tools/collect_profiles.sh.

  tools/make_traffic.py <config> <profiles/..._summary_pmc_and_kernel_stats.json> <kernel> "<workload>" [algorithmic bytes per launch]

HBM bytes per launch = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM section: unit KB;
on gfx950 FETCH_SIZE reports half of the bytes of a 16-B-per-lane streaming read, WRITE_SIZE is exact);
TCC_MISS x 128 B is written beside it as the independent line count."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    config, summary, kernel, workload = sys.argv[1:5]
    alg = float(sys.argv[5]) if len(sys.argv) > 5 else None
    pmc = json.load(open(summary))["pmc_avg_per_dispatch"][kernel]
    out = {
        "kernel": kernel,
        "workload": workload,
        "FETCH_SIZE_KB_raw": pmc["FETCH_SIZE"],
        "WRITE_SIZE_KB_raw": pmc["WRITE_SIZE"],
        "TCC_MISS_sum": pmc["TCC_MISS_sum"],
        "TCC_MISS_x128B": pmc["TCC_MISS_sum"] * 128,
        "correction": "MI355X_MICROARCH.md HBM section: unit KB (x1024); gfx950 FETCH_SIZE reports 1/2 of the bytes of a "
                      "16-B-per-lane streaming read -> x2; WRITE_SIZE exact; TCC_MISS x 128 B beside it as the line count",
        "hbm_bytes_per_launch": pmc["FETCH_SIZE"] * 2 * 1024 + pmc["WRITE_SIZE"] * 1024,
        "source": os.path.relpath(os.path.abspath(summary), ROOT) + " (rocprofv3 --pmc, separate passes, tools/collect_profiles.sh)",
    }
    if alg:
        out["algorithmic_bytes_per_launch"] = alg
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / alg
    path = os.path.join(ROOT, "profiles", "traffic_config%s.json" % config)
    json.dump(out, open(path, "w"), indent=1)
    print(path, json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
