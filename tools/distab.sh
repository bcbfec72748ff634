#!/bin/bash
# tools/distab.sh <genomes> <variant or ENV=..>...: the bench's distance stage (array resident) for ab/libskx_<variant>.so (or the built library under an
# environment setting), back to back, twice
g=${1:-1000}; shift
cp ska.rust_amd/libskx.so /tmp/distab_orig.so
for rep in 1 2; do
  for v in "$@"; do
    e=""; case "$v" in *=*) e="$v"; cp /tmp/distab_orig.so ska.rust_amd/libskx.so;; *) cp ab/libskx_$v.so ska.rust_amd/libskx.so;; esac
    env $e timeout 900 python bench.py --genomes $g --steps 1 --warmup 1 --cpu-genomes 0 --no-pmc --no-e2e --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['distance']
print('$v', {k: d[k] for k in ('wall_s','kernels_ms','pairs_per_s','first_pair')})"
  done
done
cp /tmp/distab_orig.so ska.rust_amd/libskx.so
