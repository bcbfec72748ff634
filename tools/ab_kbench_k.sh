#!/bin/bash
# tools/ab_kbench_k.sh <genomes> <k> <variant>...: as ab_kbench.sh at another k (k = 41: the 128-bit kernels)
g=${1:-500}; k=${2:-41}; shift; shift
for rep in 1 2; do
  for v in "$@"; do
    cp ab/libskx_$v.so ska.rust_amd/libskx.so
    echo -n "$v "; timeout 600 python tools/kbench.py $g 5000000 3 $k 2>/dev/null | tail -1
  done
done
