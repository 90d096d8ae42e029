#!/bin/bash
# Is the DEVICE code of the current tree the same as that of a GPU-verified commit?  Compiles every kernel source of both trees to
# gfx950 assembly (hipcc -S --cuda-device-only, the build's flags) and compares it, ignoring comments and the per-compilation-unit
# ID symbols.  Used at the end of round 4 (no GPU minutes left) to show that the emulator hooks added to the sources
# (MEDT_STATIC_SHARED, MEDT_WAVE_LOCKSTEP, vector typedefs, the asm store's C alternative) leave the GPU build untouched.
#   scripts/isa_identical.sh <git-ref>          -> one line per source file
set -u
ref=${1:?usage: isa_identical.sh <git-ref>}
root=$(cd "$(dirname "$0")/.." && pwd)
old=$(mktemp -d); o1=$(mktemp -d); o2=$(mktemp -d)
git -C "$root" archive "$ref" medical-transformer_amd/csrc include | tar -x -C "$old"
build() { # tree, outdir, file, extra flags, suffix
  local extra=""; [ "$3" = block_small ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $4 -I"$1/medical-transformer_amd/csrc" -I"$1/include" -S --cuda-device-only \
      "$1/medical-transformer_amd/csrc/$3.hip" -o "$2/$3$5.s" 2>/dev/null
}
for f in pointwise axial_core conv elementwise conv_mfma axial_small conv_small axial_stats defer axial_bwd axial_fast block_small; do
  ( [ -f "$old/medical-transformer_amd/csrc/$f.hip" ] && build "$old" "$o1" $f "" ""; build "$root" "$o2" $f "" "" ) &
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
done
( build "$old" "$o1" axial_fast "-DMEDT_FAST_BF16=1" _bf16; build "$root" "$o2" axial_fast "-DMEDT_FAST_BF16=1" _bf16 ) &
wait
strip() { grep -v '^\s*;\|__hip_cuid_\|^\s*\.\(file\|ident\)' "$1" | sed 's/;.*$//'; }
for f in "$o2"/*.s; do
  b=$(basename "$f")
  if [ ! -f "$o1/$b" ]; then echo "$b: new file"; continue; fi
  if diff -q <(strip "$o1/$b") <(strip "$f") > /dev/null; then echo "$b: device code IDENTICAL to $ref"
  else echo "$b: differs from $ref in $(diff <(strip "$o1/$b") <(strip "$f") | grep -c '^[<>]') lines (kernels whose text changed:" \
       "$(diff <(strip "$o1/$b") <(strip "$f") | grep -o '_ZN4medt[A-Za-z0-9_]*' | sort -u | head -5 | tr '\n' ' '))"; fi
done
rm -rf "$old" "$o1" "$o2"
