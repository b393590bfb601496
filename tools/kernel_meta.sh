#!/bin/bash
# Register / scratch metadata of every kernel in an object file or shared library built by the Makefile
# (VGPRs, AGPRs, spilled VGPRs, per-lane scratch bytes).  usage: tools/kernel_meta.sh build/obj/zkt_msm.o [name-filter]
set -e
F=$(readlink -f "$1"); T=$(mktemp -d); cp "$F" $T/in.o; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o > /dev/null 2>&1
for co in in.o.*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | awk '
    /\.agpr_count:/ {a=$2} /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {s=$2}
    /\.wavefront_size:/ {printf "%-90s vgpr %3d agpr %3d spill %4d scratch %6d B\n", n, v, a, s, p}'
done | c++filt | sed 's/zkt:://g' | grep -i "${2:-.}" | sort -u
rm -rf $T
