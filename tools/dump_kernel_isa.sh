#!/usr/bin/env bash
# Developer script: disassembly of one kernel of the built library.   bash tools/dump_kernel_isa.sh "<kernel substring>" > out.s
set -eu
here="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d /tmp/bvh_isa_XXXX)
cp "$here/bvh_amd/lib/libbvh_amd.so" "$tmp/lib.so"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1)
for f in "$tmp"/lib.so.*amdgcn*; do
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --demangle "$f" > "$f.s"
  n=$(grep -n -F "$1" "$f.s" | grep ">:$" | head -1 | cut -d: -f1 || true)
  if [ -n "$n" ]; then awk -v n="$n" 'NR>=n' "$f.s" | awk '/^[0-9a-f]+ <.*>:$/ && NR>1 {exit} {print}'; break; fi
done
rm -r "$tmp"
