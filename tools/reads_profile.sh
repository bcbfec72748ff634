#!/bin/bash
# tools/reads_profile.sh [tag]: rocprofv3 --kernel-trace of the read-set kernels (tools/reads_bench.py 4 50 41), per-kernel averages into gpurun_out/<tag>_reads_kernel_stats.txt
tag=${1:-reads}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fx_reads; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/fx_reads -- python $root/tools/reads_bench.py 4 50 41 > /tmp/fx_reads.log 2>&1
python - /tmp/fx_reads > $out/${tag}_reads_kernel_stats.txt <<'PY'
import csv, glob, sys, collections
t = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"].split("(")[0][:62]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("# rocprofv3 --kernel-trace of: python tools/reads_bench.py 4 50 41 (2 x 150 bp, 50 x, 252 Mbases per isolate, k = 41, --min-count 5, strict q20)")
for n, x in sorted(t.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{n:62s} calls {len(x):4d} avg {sum(x)/len(x):8.3f} ms")
PY
grep -v "^\[\|^W2\|^E2" /tmp/fx_reads.log | tail -2; cat $out/${tag}_reads_kernel_stats.txt
