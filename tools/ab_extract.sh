#!/bin/bash
# tools/ab_extract.sh <variant>...: the bench step (1 000 genomes, kernels only) once per engine build ab/libskx_<variant>.so, twice round robin
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/libskx_$v.so ska.rust_amd/libskx.so
    timeout 900 python bench.py --no-e2e --no-check --no-distance --cpu-genomes 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['stage_ms_per_step'].items() if x}, 'frac', round(d['roofline']['frac'],4), d['config'].get('rows_U'), d['config'].get('rows_kept'))"
  done
done
