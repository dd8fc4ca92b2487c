#!/bin/bash
# Developer tool: builds libspal_hip.so with extra -D flags for spal_coo.hip / spal_csr.hip into
# spalinalg_amd/lib_var/<name>/ (select at run time with SPAL_HIP_LIB=<path>).
# usage: tools/build_variant.sh <name> <file.hip> "<flags>"
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
make -s -C spalinalg_amd/csrc
out=spalinalg_amd/lib_var/$name
mkdir -p "$out"
HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread $HIPFLAGS $flags -c spalinalg_amd/csrc/$src -o "$out/$src.o"
objs=""
for o in spalinalg_amd/lib/obj/*.o; do
  case "$o" in */$src.o) objs="$objs $out/$src.o" ;; *) objs="$objs $o" ;; esac
done
/opt/rocm/bin/hipcc -shared -fPIC $HIPFLAGS -o "$out/libspal_hip.so" $objs -pthread -ldl
echo "$out/libspal_hip.so"
