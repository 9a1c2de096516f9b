# Run ON THE GPU BOX: expand_hits_kernel's average duration per library build (rocprofv3 kernel stats, config 3 at 4 GiB)
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  rm -rf /tmp/eh
  ACM_NATIVE_LIB=$GRAFT_REPO_ROOT/aho-corasick-1975_amd/$l rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eh -- python3 $GRAFT_REPO_ROOT/bench.py --config 3 --mib 4096 --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/eh/*/*kernel_stats.csv")[0])):
    if "expand_hits" in r["Name"] or "scan_gram_kernel<false" in r["Name"]: print("$l", r["Name"][27:60], r["AverageNs"])
PY
done
