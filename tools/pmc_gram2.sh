#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the two 4-gram kernels side by side (ACM_GPU_GRAM2=0 / 1), config 3's
# dictionary on 2 GiB, one launch each of the record and the count-only kernel per pass.
#   tools/pmc_gram2.sh <outdir under gpurun_out> [lib suffix]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
[ -n "$2" ] && export ACM_NATIVE_LIB=$GRAFT_REPO_ROOT/aho-corasick-1975_amd/libac75_amd_$2.so
for v in 0 1; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"; do
    i=$((i+1)); rm -rf /tmp/pg$v$i
    ACM_GPU_GRAM2=$v timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "scan_gram" --output-format csv -d /tmp/pg$v$i -- python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 2048 > /dev/null 2>&1 || echo "pass $v $i failed"
  done
done
python3 - <<PY > $OUT/pmc_gram2.txt
import csv,glob,collections
for v in (0, 1):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (1, 2):
        for f in glob.glob("/tmp/pg%d%d/*/*counter_collection.csv" % (v, d)):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in agg.items():
        print("ACM_GPU_GRAM2=%d" % v, k)
        print("   ", {n: round(sum(x)/len(x)) for n,x in c.items()}, "launches", max(len(x) for x in c.values()))
PY
cat $OUT/pmc_gram2.txt
