#!/bin/bash
# tools/ab_reads.sh <variant>...: per-kernel averages of the read-set kernels (rocprofv3 --kernel-trace, tools/reads_bench.py 3 isolates, k = ${K:-41})
# for ab/libskx_<variant>.so, back to back on one box.  CMD="python bench.py ..." tools/ab_reads.sh <variant>...: the same table for another command
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out; k=${K:-41}
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  cp $root/ab/libskx_$v.so $root/ska.rust_amd/libskx.so
  rm -rf /tmp/abr_$v; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/abr_$v -- ${CMD:-python $root/tools/reads_bench.py 3 50 $k} > /tmp/abr_$v.log 2>&1
  tail -1 /tmp/abr_$v.log | cut -c1-120
  python - $v /tmp/abr_$v <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1:3]
t = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
rows = sorted(t.items(), key=lambda kv: -sum(kv[1]))[:7]
print(v, " | ".join(f"{n.replace('void skx::','')} {sum(x)/len(x):.3f}x{len(x)}" for n, x in rows))
PY
done
