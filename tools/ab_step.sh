#!/bin/bash
# tools/ab_step.sh <genomes> <variant>...: the device-resident step only (bench.py without the executable, the oracle check and the distance stage),
# once per engine build ab/libskx_<variant>.so, twice round robin on one box
g=${1:-1000}; shift
for rep in 1 2; do
  for v in "$@"; do
    cp ab/libskx_$v.so ska.rust_amd/libskx.so
    timeout 600 python bench.py -k ${K:-31} --genomes $g --steps 5 --warmup 2 --cpu-genomes 0 --no-pmc --no-e2e --no-check --no-distance 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['stage_ms_per_step'].items() if x}, d['config']['rows_U'], d['config']['rows_kept'])"
  done
done
