#!/bin/bash
# Developer tool: builds libspal_hip.so with extra -D flags for some of its sources into
# spalinalg_amd/lib_var/<name>/ (select at run time with SPAL_HIP_LIB=<path>).
# usage: tools/build_variant.sh <name> "<file.hip> [<file.hip> ...]" "<flags>"
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; srcs=$2; flags=$3
make -s -C spalinalg_amd/csrc -j8
out=spalinalg_amd/lib_var/$name
mkdir -p "$out"
HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics"
for src in $srcs; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread $HIPFLAGS $flags -c spalinalg_amd/csrc/$src -o "$out/$src.o" &
done
wait
objs=""
for o in spalinalg_amd/lib/obj/*.o; do
  b=$(basename "$o" .o)
  if [[ " $srcs " == *" $b "* ]]; then objs="$objs $out/$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC $HIPFLAGS -o "$out/libspal_hip.so" $objs -pthread -ldl
echo "$out/libspal_hip.so"
