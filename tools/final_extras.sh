#!/bin/bash
# tools/final_extras.sh <tag>: the rest of the evidence of the tree as it is (gpurun -- 'bash tools/final_extras.sh r03_final'):
#   gpurun_out/<tag>_bench_full.json            the default bench line (what the driver runs), stderr beside it
#   gpurun_out/<tag>_reads_kernel_stats.txt     rocprofv3 --kernel-trace of tools/reads_bench.py 4 50 41 (the read-set kernels per isolate)
#   gpurun_out/<tag>_distance_kernel_stats.txt  rocprofv3 --kernel-trace of the bench's distance stage (1 000 samples)
tag=${1:-final}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out
python $root/bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err
summ() {   # <dir> <header> : per-kernel averages of a rocprofv3 csv trace
python - "$1" "$2" <<'PY'
import csv, glob, sys, collections
d, head = sys.argv[1:3]
t = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"].split("(")[0][:62]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("# " + head)
for n, x in sorted(t.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{n:62s} calls {len(x):4d} avg {sum(x)/len(x):8.3f} ms")
PY
}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fx_reads; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/fx_reads -- python $root/tools/reads_bench.py 4 50 41 > /tmp/fx_reads.log 2>&1
summ /tmp/fx_reads "rocprofv3 --kernel-trace of: python tools/reads_bench.py 4 50 41 (2 x 150 bp, 50 x, 252 Mbases per isolate, k = 41, --min-count 5, strict q20); $(tail -1 /tmp/fx_reads.log | cut -c1-110)" > $out/${tag}_reads_kernel_stats.txt
rm -rf /tmp/fx_dist; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/fx_dist -- python $root/bench.py --genomes 1000 --steps 1 --warmup 1 --cpu-genomes 0 --no-pmc --no-e2e --no-check > /tmp/fx_dist.log 2>&1
summ /tmp/fx_dist "rocprofv3 --kernel-trace of: python bench.py --genomes 1000 --steps 1 --warmup 1 --no-e2e --no-check (the step + ska distance on the resident array: 499 500 pairs x 22.5 M rows)" > $out/${tag}_distance_kernel_stats.txt
