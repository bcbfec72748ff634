#!/bin/bash
# tools/ab_sort.sh <variant>...: per-kernel averages of the radix sort's kernels (rocprofv3 --kernel-trace of skx_debug_prims_selftest at ${N:-40000000} keys)
# for ab/libskx_<variant>.so, back to back on one box
root=$(cd "$(dirname "$0")/.." && pwd); n=${N:-40000000}
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  cp $root/ab/libskx_$v.so $root/ska.rust_amd/libskx.so
  rm -rf /tmp/abs_$v; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/abs_$v -- python -c "
import ctypes, sys
L = ctypes.CDLL('$root/ska.rust_amd/libskx.so')
L.skx_debug_prims_selftest.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]
print('selftest', L.skx_debug_prims_selftest(0, $n, 5))
" > /tmp/abs_$v.log 2>&1
  grep selftest /tmp/abs_$v.log
  python - $v /tmp/abs_$v <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1:3]
t = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
rows = sorted(t.items(), key=lambda kv: -sum(kv[1]))[:6]
print(v, " | ".join(f"{n.replace('void skx::','')} {sum(x)/len(x):.3f}x{len(x)}" for n, x in rows))
PY
done
