#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace summary and PMC passes of a bench command.
#   tools/collect_profiles.sh <tag> [bench.py arguments, e.g. --config 3 --steps 3 --warmup 1]
# Writes under gpurun_out/profile_<tag>/; copy what should be judged into profiles/.
set -e
R=${1:-r03}
shift || true
ARGS="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --no-other-configs --no-multi-c-abi > $OUT/bench_under_trace.json 2> $OUT/trace.err
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr " " "_" | cut -c1-32)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-other-configs --no-multi-c-abi > /dev/null 2> $OUT/pmc_$name.err || echo "pmc $name failed"
  echo "pass $name done" >> $OUT/progress.txt
done
python3 - <<PY
import csv, glob, collections, json, re
out = {"command": "bench.py $ARGS"}
f = glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
out["kernel_stats"] = [dict(name=r["Name"][:110], calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"])) for r in csv.DictReader(open(f))][:14]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = next((x for x in ("scan_dense_kernel", "scan_gram2_kernel", "scan_gram_kernel", "scan_short_kernel", "scan_starts_kernel", "expand_items", "expand_hits", "close_holes", "tile_gather", "tile_size", "order_bucket", "order_finish") if x in n), None)
        if k == "scan_gram_kernel" and re.search(r"scan_gram_kernel<[^>]*, true>", n):
            k = "scan_gram_kernel_tiled"   # (the e2e leg's scan: acm_gpu_scan_ordered_device)
        # record-mode instantiations only (the count-only pass that sizes the record buffer is another kernel)
        if k == "scan_gram2_kernel" and re.search(r"scan_gram2_kernel<[^>]*, true>", n):
            k = "scan_gram2_kernel_tiled"
        count_only = ("scan_gram_kernel<true" in n or "scan_gram2_kernel<true" in n or "scan_short_kernel<true" in n or
                      (("scan_dense_kernel" in n or "scan_starts_kernel" in n) and ", true>(" in n))
        if k and not count_only:
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_avg_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out["pmc_avg_per_dispatch"], indent=1))
for r in out["kernel_stats"][:6]:
    print(r)
PY
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
