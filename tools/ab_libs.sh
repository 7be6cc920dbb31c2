#!/bin/bash
# A/B harness: run a command once per candidate library build/ab/lib_<tag>.so (copied over the in-tree .so on the box).
# Usage: ab_libs.sh "<tags>" <command...>
tags=$1; shift
for t in $tags; do
  cp build/ab/lib_$t.so stylemesh_amd/libstylemesh_hip.so
  echo "=== $t"
  "$@"
done
