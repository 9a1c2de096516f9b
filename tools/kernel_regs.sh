#!/bin/bash
# Register use and spills of the scan kernels for a set of -D switches (device-only compile, no link):
#   tools/kernel_regs.sh [-DACM_GRAM_PUSH2=1 ...]   -> name, sgprs, sgpr spills, vgprs, vgpr spills, scratch bytes
cd "$(dirname "$0")/../aho-corasick-1975_amd/csrc" || exit 1
out=/tmp/kregs_$$.co
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -I../../include -I. "$@" -c -o $out acm_gpu.hip 2>/dev/null || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out | awk '
  /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2} /\.sgpr_count:/ {s=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {vs=$2; if (n ~ /scan_(gram2?|dense|starts|short)_kernel/) print n, "sgpr", s, "spill", ss, "vgpr", v, "spill", vs, "scratch", p}' | sed 's/_ZN12_GLOBAL__N_1[0-9]*//; s/EEvNS.*E[a-z]* / /' | sort
rm -f $out
