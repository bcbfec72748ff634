#!/bin/bash
# A/B timing of engine builds on one GPU box: tools/ab.sh <genomes> <variant>...   (variants = ab/libskx_<variant>.so)
# Every variant runs the same bench.py workload back to back, twice, so box-to-box and run-to-run noise is visible.
g=${1:-1000}; shift
for rep in 1 2; do
  for v in "$@"; do
    cp ab/libskx_$v.so ska.rust_amd/libskx.so
    timeout 600 python bench.py --genomes $g --steps 3 --warmup 1 --cpu-genomes 0 --no-pmc 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['stage_ms_per_step'].items() if x}, d['config']['rows_U'], d['config']['rows_kept'])"
  done
done
