#!/bin/bash
# quick look: bench numbers + L2 miss/fetch counters of the scan kernel
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"; done
rm -rf /tmp/qp; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/qp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/qp2; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/qp2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("/tmp/qp","/tmp/qp2"):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "scan_dense" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, sum(v)/len(v))
PY
