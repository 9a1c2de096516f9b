# Run ON THE GPU BOX: a few PMC passes over config 3's kernels at 4 GiB (quick look, not the judged profile).
#   tools/quick_pmc_c3.sh <outdir under gpurun_out> "<counter set>" ["<counter set>" ...]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY"; fi
for set in "$@"; do
  name=$(echo $set | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --mib 4096 --steps 2 --warmup 1 --prewarm-ms 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$name.err || echo fail $name
done
python3 - <<PY
import csv, glob, collections
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "scan_gram_kernel<false" in n: pmc["gram"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "expand_hits" in n: pmc["expand"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in pmc.items():
    print(k, {c: "%.4g" % (sum(v)/len(v)) for c, v in d.items()})
PY
