#!/bin/bash
# Build container: the gfx950 ISA and the register / LDS / scratch figures of one kernel of a
# compiled object (a COPY of the object is taken apart: llvm-objcopy rewrites its input).
#   usage: profiles/kernel_isa.sh <object under csrc/build> <mangled-name substring> [out.s]
set -e
OBJ=$1; SYM=$2; OUT=${3:-/tmp/kernel_isa.s}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
cp "$OBJ" "$T/in.o"
$B/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$T/in.o" "$T/out.o"
$B/clang-offload-bundler --unbundle --type=o --input="$T/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/dev.co"
$B/llvm-readelf --notes "$T/dev.co" | grep -B2 -A40 "\.name:.*$SYM" | grep -E "\.name:|vgpr_count|sgpr_count|spill_count|private_segment_fixed|group_segment_fixed" 
$B/llvm-objdump -d "$T/dev.co" > "$T/all.s"
awk -v s="$SYM" '$0 ~ "^[0-9a-f]+ <.*" s ".*>:" {on=1} on {print} on && /s_endpgm/ {exit}' "$T/all.s" > "$OUT"
echo "$(wc -l < "$OUT") lines -> $OUT"
rm -rf "$T"
