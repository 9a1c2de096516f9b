#!/bin/bash
# Run ON THE GPU BOX: scan_gram2_kernel's ablation builds (make expd D=ACM_GRAM2_ABLATE=n) on 2 GiB of config 3.
#   tools/ablate_gram2.sh <outdir under gpurun_out> n n n ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
{
for a in "$@"; do
  echo "== ACM_GRAM2_ABLATE=$a"
  ACM_NATIVE_LIB=$GRAFT_REPO_ROOT/aho-corasick-1975_amd/libac75_amd_ACM_GRAM2_ABLATE=$a.so timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 2048 2>&1 | grep -v amdgpu.ids
done
echo "== product"
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 2048 2>&1 | grep -v amdgpu.ids
echo "== product, ACM_GPU_GRAM2=0"
ACM_GPU_GRAM2=0 timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/exp_c3.py 2048 2>&1 | grep -v amdgpu.ids
} > $OUT/ablation.txt 2>&1
cat $OUT/ablation.txt
