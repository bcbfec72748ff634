#!/bin/bash
# tools/ab_env.sh <VAR>: the bench step (1 000 genomes, kernels only) with and without VAR=1 in the environment, twice round robin
for rep in 1 2; do
  for v in "" "$1=1"; do
    env $v timeout 900 python bench.py --no-e2e --no-check --no-distance --cpu-genomes 0 --no-pmc --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['stage_ms_per_step'].items() if x}, d['config'].get('rows_U'), d['config'].get('rows_kept'))"
  done
done
