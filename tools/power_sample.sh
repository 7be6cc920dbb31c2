#!/bin/bash
# Samples rocm-smi power / clocks while a command runs (run on the GPU box). Usage: power_sample.sh <out.txt> <command...>
out=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
: > $out
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use|mclk" | tr -s ' ' | tr '\n' ';' >> $out
  echo >> $out
  sleep 0.2
done
wait $pid
