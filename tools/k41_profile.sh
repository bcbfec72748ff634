#!/bin/bash
# tools/k41_profile.sh <tag>: the bench step at k = 41 (500 genomes: the 128-bit kernels) + the rocprofv3 kernel summary of the same command
tag=${1:-k41}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out
B="python $root/bench.py --no-e2e --no-check --no-distance --cpu-genomes 0 --no-pmc --genomes 500 -k 41 --steps 5 --warmup 2"
$B 2>/dev/null | tail -1 > $out/${tag}_bench_k41.json
python -c "
import json; d=json.load(open('$out/${tag}_bench_k41.json')); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if v}, 'frac', round(d['roofline']['frac'],4))"
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_k41trace; (cd $root && rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_k41trace -- $B > /dev/null 2>&1)
python - "$out/${tag}_k41trace" > $out/${tag}_kernel_stats_k41.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in acc.values()) or 1
print(f"{'kernel':<100} {'calls':>6} {'total_ms':>11} {'avg_ms':>10} {'min_ms':>10} {'max_ms':>10} {'pct':>6}")
for k, a in sorted(acc.items(), key=lambda x: -x[1][1])[:12]:
    print(f"{k[:100]:<100} {a[0]:>6} {a[1]:>11.3f} {a[1] / a[0]:>10.3f} {a[2]:>10.3f} {a[3]:>10.3f} {100 * a[1] / tot:>6.2f}")
PY
cat $out/${tag}_kernel_stats_k41.txt
