#!/bin/bash
# tools/pmc.sh "<cmd>" COUNTER...   -- one rocprofv3 pass per counter (gpurun refuses --pmc with trace domains other than kernel-trace),
# per-kernel mean per launch printed at the end.  Run on the GPU box: gpurun -- 'tools/pmc.sh "python tools/kbench.py 200 5000000 2" SQ_WAVES ...'
cmd=$1; shift
out=gpurun_out/pmc_$$; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  (cd $root && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/$out/$c -- $cmd > $root/$out/$c.log 2>&1)
done
cd $root
python - "$out" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in sys.argv[2:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0][:60]
            acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(acc.items(), key=lambda x: -x[1][1])[:8]:
        print(f"{c:28s} {k:62s} launches={n} mean={v / n:.4g}")
PY
