#!/bin/bash
# Run ON THE GPU BOX: the round-4 evidence beside the bench profiles -- scan_gram2_kernel's ablation
# builds, its cycle stamps (diag build), both 4-gram kernels side by side, the short-keyword passes.
# Needs: make -C aho-corasick-1975_amd/csrc expd D=ACM_GRAM2_ABLATE={7,6,5,4,3,2,1}; make diag.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04_extras}
mkdir -p $O
echo "start" > $O/progress.txt
$R/tools/ablate_gram2.sh ${1:-r04_extras}/ablate 7 6 5 4 3 2 1 > /dev/null 2>&1 || exit 1
echo "ablation done" >> $O/progress.txt
ACM_NATIVE_LIB=$R/aho-corasick-1975_amd/libac75_amd_diag.so timeout -k 10 300 python3 $R/tools/diag_gram2.py 2>&1 | grep -v amdgpu.ids > $O/cycle_stamps.txt || exit 1
echo "stamps done" >> $O/progress.txt
timeout -k 10 300 python3 $R/tools/exp_gram2.py 2048 2>&1 | grep -v amdgpu.ids > $O/gram2_vs_gram.txt || exit 1
echo "exp_gram2 done" >> $O/progress.txt
$R/tools/ablate_short.sh ${1:-r04_extras}/short 0 || exit 1
echo "short done" >> $O/progress.txt
