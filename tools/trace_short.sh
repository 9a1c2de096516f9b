#!/bin/bash
# Run ON THE GPU BOX: kernel trace of tools/exp_short.py (which pass of a short-keyword dictionary takes what).
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace_short}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/exp_short.py 2048 > $out/trace.log 2>&1 || exit 1
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cut -d, -f1-6 "$f" | head -12 > $out/kernel_stats_head.txt
