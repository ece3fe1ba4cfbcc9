#!/bin/bash
# dev: registers / spills / LDS of every kernel of a built library (default: the shipped one): tools/kernel_regs.sh [lib.so] [name filter]
LIB=${1:-pointnerf_amd/libpnerf_hip.so}; F=${2:-.}
T=$(mktemp -d); cp $LIB $T/lib.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1)
for co in $T/lib.so.*gfx950; do
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | awk '
  /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.vgpr_spill_count:/ {sp=$2} /\.sgpr_count:/ {sg=$2} /\.group_segment_fixed_size:/ {l=$2}
  /\.wavefront_size:/ {printf "%-100s vgpr %3s agpr %3s sgpr %3s spill %3s lds %6s\n", name, v, a, sg, sp, l}'
done | grep -E "$F" | sort
rm -rf $T
