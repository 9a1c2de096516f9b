#!/bin/bash
# Register use and spills of ONE kernel family, compiled alone (seconds): dev_all.h + the instantiations given.
#   tools/kernel_regs_one.sh 'scan_gram2_kernel<false,false>' 'scan_gram2_kernel<false,true>' [-- -DACM_... ...]
# Prints name, sgprs, sgpr spills, vgprs, vgpr spills, scratch bytes, LDS; keeps the ISA in /tmp/kregs_one.s
cd "$(dirname "$0")/../aho-corasick-1975_amd/csrc" || exit 1
src=/tmp/kregs_one_$$.hip
out=/tmp/kregs_one_$$.co
{
  echo '#include <hip/hip_runtime.h>'
  echo '#include <cstdio>'
  echo '#include "dev_all.h"'
  i=0
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do
    echo "const void *acm_inst_$i = reinterpret_cast<const void *> (&$1);"
    i=$((i + 1)); shift
  done
} > $src
[ "$1" = "--" ] && shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -I../../include -I. "$@" -c -o $out $src || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -I../../include -I. "$@" -S -o /tmp/kregs_one.s $src 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out | awk '
  /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2} /\.sgpr_count:/ {s=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {vs=$2; print n, "sgpr", s, "spill", ss, "vgpr", v, "spill", vs, "scratch", p}' | sed 's/_ZN12_GLOBAL__N_1[0-9]*//; s/EEvNS.*E[a-z]* / /' | sort
rm -f $src $out
