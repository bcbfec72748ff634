#!/bin/bash
# tools/envab.sh <genomes> "<ENV=1 ...>" "<ENV=0 ...>" ...: the bench workload under different environments (run-time switches), back to back, twice
g=${1:-1000}; shift
for rep in 1 2; do
  for v in "$@"; do
    env $v timeout 600 python bench.py --genomes $g --steps 3 --warmup 1 --cpu-genomes 0 --no-pmc --no-e2e --no-distance 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['stage_ms_per_step'].items() if x}, d['config']['rows_U'], d['config']['rows_kept'], d.get('check'))"
  done
done
