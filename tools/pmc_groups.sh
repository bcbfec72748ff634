#!/bin/bash
# tools/pmc_groups.sh "<cmd>" "C1 C2 C3" "C4 C5" ...   -- one rocprofv3 pass per GROUP of counters (a group must fit the SQ's counter
# slots; --kernel-trace is the only trace domain, as gpurun requires), per-kernel mean per launch printed at the end.
cmd=$1; shift
out=gpurun_out/pmcg_$$; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i + 1))
  (cd $root && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/$out/g$i -- $cmd > $root/$out/g$i.log 2>&1)
done
cd $root
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:70]
        a = acc[(row["Counter_Name"], k)]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for (c, k), (n, v) in sorted(acc.items(), key=lambda x: (x[0][1], x[0][0])):
    if v / n >= 1e5: print(f"{k:72s} {c:30s} launches={n} mean={v / n:.4g}")
PY
