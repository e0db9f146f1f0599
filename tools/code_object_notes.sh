#!/bin/bash
# Registers, spills and scratch of every kernel of the shipped library (code-object notes of the gfx950 bundle).
#   tools/code_object_notes.sh > profiles/rNN_code_object_notes.txt
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
cp playsnark_amd/libplaysnark_hip.so "$tmp/lib.so"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
co=$(ls "$tmp"/lib.so.*gfx950 | head -1)
echo "# kernel  vgpr  agpr  spilled_vgpr  scratch_bytes   ($(git rev-parse --short HEAD), $(basename "$co"))"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | grep -E "\.name:|\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.agpr_count" | paste - - - - - |
  sed -E 's/.*agpr_count: *([0-9]+).*\.name: *([^ \t]+).*fixed_size: *([0-9]+).*vgpr_count: *([0-9]+).*spill_count: *([0-9]+).*/\2 \4 \1 \5 \3/' |
  while read name v a s p; do echo "$(echo "$name" | c++filt | sed -E 's/\(.*//; s/^void //') $v $a $s $p"; done | sort
rm -rf "$tmp"
