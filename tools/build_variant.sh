#!/bin/bash
# Builds build/ab/lib_<tag>.so: the in-tree library with <file>.hip (default conv) recompiled under extra flags.
# Usage: build_variant.sh <tag> [-f file] <extra hipcc flags...>      (then: tools/ab_libs.sh "<tags>" <command>)
set -euo pipefail
cd "$(dirname "$0")/.."
tag=$1; shift
file=conv
if [ "${1:-}" = "-f" ]; then file=$2; shift 2; fi
bash stylemesh_amd/csrc/build.sh >/dev/null
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function "$@" \
  -c stylemesh_amd/csrc/$file.hip -o build/ab/${file}_$tag.o
objs=()
for f in conv texture gram prep comm eval raster scatter_plan exchange replay; do
  if [ $f = $file ]; then objs+=(build/ab/${file}_$tag.o); else objs+=(build/$f.o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/lib_$tag.so "${objs[@]}" -L/opt/rocm/lib -lrccl
echo "built build/ab/lib_$tag.so"
