#!/bin/bash
# A/B of library builds under rocprofv3: args = suffixes of aho-corasick-1975_amd/libac75_amd_<sfx>.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  rm -rf /tmp/ab_$v
  ACM_NATIVE_LIB=$GRAFT_REPO_ROOT/aho-corasick-1975_amd/libac75_amd_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v step', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_ms'])"
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ab_$v/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "scan_dense" in r["Name"] or "expand" in r["Name"]: print("   ", r["Name"][27:48], r["AverageNs"])
PY
done; done
