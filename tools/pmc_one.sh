#!/bin/bash
# Run ON THE GPU BOX: the given counter sets (one rocprofv3 pass each, ~1 min) over the 4-gram kernels of
# `tools/exp_c3.py 2048`; progress goes to gpurun_out/<outdir>/progress.txt.
#   [ACM_GPU_GRAM2=0] tools/pmc_one.sh <outdir under gpurun_out> "<set>" ["<set>" ...]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/po$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "scan_gram" --output-format csv -d /tmp/po$i -- python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 2048 > /tmp/po$i.log 2>&1 || { echo "pass $i ($set) failed" >> $OUT/progress.txt; tail -3 /tmp/po$i.log >> $OUT/progress.txt; }
  echo "pass $i done" >> $OUT/progress.txt
done
python3 - <<PY > $OUT/pmc.txt
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/po*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    print(k)
    print("   ", {n: round(sum(x)/len(x)) for n,x in c.items()}, "launches", max(len(x) for x in c.values()))
PY
cat $OUT/pmc.txt
