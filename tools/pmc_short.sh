#!/bin/bash
# Run ON THE GPU BOX: the given counter sets (one rocprofv3 pass each) over the kernels of
# `tools/exp_short.py 2048` (scan_gram2_kernel + scan_short_kernel + close_holes_kernel); progress goes to
# gpurun_out/<outdir>/progress.txt.    tools/pmc_short.sh <outdir under gpurun_out> "<set>" ["<set>" ...]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/ps$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "scan_|close_holes" --output-format csv -d /tmp/ps$i -- python3 $GRAFT_REPO_ROOT/tools/exp_short.py 2048 > /tmp/ps$i.log 2>&1 || { echo "pass $i ($set) failed" >> $OUT/progress.txt; tail -3 /tmp/ps$i.log >> $OUT/progress.txt; }
  echo "pass $i done" >> $OUT/progress.txt
done
python3 - <<PY > $OUT/pmc.txt
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ps*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(scan_\w+<[^>]*>|close_holes_kernel)", n)
        agg[m.group(1) if m else n[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in sorted(agg.items()):
    print(k)
    print("   ", {n: round(sum(x)/len(x)) for n,x in sorted(c.items())}, "launches", max(len(x) for x in c.values()))
PY
cat $OUT/pmc.txt
