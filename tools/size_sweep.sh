for cfg in "4000 1000000" "2000 2000000" "1300 3000000" "1000 4000000" "1000 5000000" "800 6000000" "600 8000000" "480 10000000"; do set -- $cfg; timeout 300 python bench.py --genomes $1 --genome-len $2 --steps 3 --warmup 1 --cpu-genomes 0 --no-e2e --no-check --no-distance 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); gb=$1*$2/1e9
print('$1 x $2:', round(d['ms_per_step'],1), 'ms', round(d['value']), 'genomes/s', round(gb/d['ms_per_step']*1e3,1), 'Gbases/s', {k: round(x,1) for k,x in d['stage_ms_per_step'].items() if x})"; done
