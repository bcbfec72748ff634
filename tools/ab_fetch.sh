#!/bin/bash
# tools/ab_fetch.sh <variant>...: per engine build ab/libskx_<variant>.so the append kernel's time (bench step, 1 000 genomes) and its FETCH_SIZE
# (one rocprofv3 --pmc pass of the same command): what a change does to the re-fetch of the regions by their A readers
root=$(pwd)
B="python $root/bench.py ${K:+-k $K} --genomes ${G:-1000} --steps 3 --warmup 1 --cpu-genomes 0 --no-pmc --no-e2e --no-check --no-distance"
for v in "$@"; do
  cp ab/libskx_$v.so ska.rust_amd/libskx.so
  ms=$($B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_ms_per_step']['append'],2), round(d['ms_per_step'],2), d['config']['rows_U'])")
  rm -rf /tmp/abf_$v; (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/abf_$v -- $B > /dev/null 2>&1)
  gb=$(python - /tmp/abf_$v <<'PY'
import csv, glob, sys
n = 0; v = 0.0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE" and ("append_kernel<false" in r["Kernel_Name"] or "append_wide_kernel<false" in r["Kernel_Name"]):
            n += 1; v += float(r["Counter_Value"])
print(round(v / max(n, 1) * 2 * 1024 / 1e9, 2))
PY
)
  echo "$v: append ms / step ms / rows = $ms ; FETCH x 2 = $gb GB per launch"
done
