#!/bin/bash
# Run ON THE GPU BOX: cache counters of the 4-gram kernels (ACM_GPU_GRAM2=0 / 1), config 3's dictionary, 2 GiB.
#   tools/pmc_gram2_mem.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
for v in 0 1; do
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
    i=$((i+1)); rm -rf /tmp/pm$v$i
    ACM_GPU_GRAM2=$v timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "scan_gram" --output-format csv -d /tmp/pm$v$i -- python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 2048 > /tmp/pm$v$i.log 2>&1 || { echo "pass $v $i ($set) failed"; tail -3 /tmp/pm$v$i.log; }
    echo "pass $v $i done" >> $OUT/progress.txt
  done
done
python3 - <<PY > $OUT/pmc_gram2_mem.txt
import csv,glob,collections
for v in (0, 1):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for d in range(1, 5):
        for f in glob.glob("/tmp/pm%d%d/*/*counter_collection.csv" % (v, d)):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in agg.items():
        print("ACM_GPU_GRAM2=%d" % v, k)
        print("   ", {n: round(sum(x)/len(x)) for n,x in c.items()}, "launches", max(len(x) for x in c.values()))
PY
cat $OUT/pmc_gram2_mem.txt
