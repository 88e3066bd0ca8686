#!/bin/bash
# Registers / scratch / LDS of every kernel of one object file (the device code object inside its .hip_fatbin section).
# usage: tools/kres.sh sibelia_amd/lib/obj/commit.o [name filter]
set -e
L=/opt/rocm/lib/llvm/bin
d=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$d/fat.bin "$1"
$L/clang-offload-bundler --unbundle --type=o --input=$d/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$d/dev.o
$L/llvm-readelf --notes $d/dev.o | grep -E "\.name:|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size" | paste - - - - - | \
  awk '{for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1);if($i==".vgpr_count:")v=$(i+1);if($i==".sgpr_count:")s=$(i+1);if($i==".private_segment_fixed_size:")p=$(i+1);if($i==".group_segment_fixed_size:")g=$(i+1)} printf "%-60s vgpr %4s sgpr %4s scratch %5s lds %6s\n", n, v, s, p, g}' | grep -E "${2:-.}"
rm -rf $d
