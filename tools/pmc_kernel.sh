#!/bin/bash
# counters of ONE kernel (regex $1) in `python3 bench.py $2...`; separate passes, no tracing beside --pmc
re=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/pk$i -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > /dev/null 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in range(1,7):
    for f in glob.glob("/tmp/pk%d/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    print("   ", {c: round(sum(x)/len(x)) for c,x in v.items()}, "launches", max(len(x) for x in v.values()))
PY
