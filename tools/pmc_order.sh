#!/bin/bash
# instruction and wait counters of the order kernels in the e2e step of config 2 (tools/exp_e2e.py); separate passes
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/po$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d /tmp/po$i -- python3 $GRAFT_REPO_ROOT/tools/exp_e2e.py ${1:-2} > /dev/null 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in range(1,5):
    for f in glob.glob("/tmp/po%d/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            nm=r["Kernel_Name"]
            if "order_" in nm or "expand" in nm:
                agg[nm[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    print("   ", {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
