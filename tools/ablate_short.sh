#!/bin/bash
# Run ON THE GPU BOX: scan_short_kernel's ablation builds (make expd D=ACM_SHORT_ABLATE=n) on
# tools/exp_short.py's dictionary, kernel times from a trace of each.   tools/ablate_short.sh <out> <n> ...
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-ablate_short}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  lib=libac75_amd.so; [ "$a" != 0 ] && lib="libac75_amd_ACM_SHORT_ABLATE=$a.so"; [ -f "$GRAFT_REPO_ROOT/aho-corasick-1975_amd/libac75_amd_$a.so" ] && lib="libac75_amd_$a.so"
  export ACM_NATIVE_LIB=$GRAFT_REPO_ROOT/aho-corasick-1975_amd/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$a -- python3 $GRAFT_REPO_ROOT/tools/exp_short.py 2048 > $out/trace$a.log 2>&1 || exit 1
  echo "== ACM_SHORT_ABLATE=$a" >> $out/ablate.txt
  grep -v amdgpu.ids $out/trace$a.log | grep "^kernel" >> $out/ablate.txt
  python3 - "$out/trace$a" >> $out/ablate.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'scan_' in r['Name'] or 'close_holes' in r['Name']:
        print('   %-50s calls %s avg %.1f us' % (r['Name'].split('(anonymous namespace)::')[1][:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
