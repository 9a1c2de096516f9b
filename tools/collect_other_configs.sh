#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel-trace summaries of the config 3 and config 5 experiment drivers.
set -e
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_${R}_configs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 1024 > $OUT/c3.log 2> $OUT/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 $GRAFT_REPO_ROOT/tools/exp_c5.py 4096 > $OUT/c5.log 2> $OUT/c5.err
cp $OUT/c3/*/*kernel_stats.csv $OUT/config3_kernel_stats.csv
cp $OUT/c5/*/*kernel_stats.csv $OUT/config5_kernel_stats.csv
grep -h "kernel=" $OUT/c3.log $OUT/c5.log
