#!/bin/bash
# tools/ab_kbench.sh <genomes> <variant>...: dictionary-build kernels only (tools/kbench.py) once per engine build ab/libskx_<variant>.so, twice round robin
g=${1:-1000}; shift
for rep in 1 2; do
  for v in "$@"; do
    cp ab/libskx_$v.so ska.rust_amd/libskx.so
    echo -n "$v "; timeout 600 python tools/kbench.py $g 5000000 3 31 2>/dev/null | tail -1
  done
done
