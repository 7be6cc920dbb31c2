#!/bin/bash
# Builds libstylemesh_hip.so for gfx950 (cross-compiles without a GPU). Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libstylemesh_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function"
mkdir -p ../../build
objs=()
for f in conv texture gram prep comm eval raster scatter_plan exchange replay; do
  o=../../build/$f.o
  stale=0
  for dep in "$f.hip" *.h ../../include/stylemesh_hip.h; do
    if [ ! -f "$o" ] || [ "$dep" -nt "$o" ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $o &
  fi
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT "${objs[@]}" -L/opt/rocm/lib -lrccl
echo "built $(realpath $OUT)"
