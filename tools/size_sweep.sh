#!/bin/bash
# tools/size_sweep.sh [k]: extraction (+ the regions' layout) per base over sample sizes, tools/kbench.py at a constant 1 Gbase per build:
# 200 x 5 Mbp, 50 x 20 Mbp, 25 x 40 Mbp, 10 x 100 Mbp, 4 x 250 Mbp -- the time per base must not depend on the sample's length
k=${1:-31}
for spec in "200 5000000" "50 20000000" "25 40000000" "10 100000000" "4 250000000"; do
  set -- $spec
  out=$(timeout 900 python tools/kbench.py $1 $2 3 $k 2>&1 | tail -1)
  python - "$1" "$2" "$out" <<'PY'
import ast, sys
n, glen, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
try:
    t = ast.literal_eval(out)
    ms = sum(v for k, v in t.items() if k in ("scatter", "hist", "dedupe"))
    print(f"{n:4d} x {glen / 1e6:6.0f} Mbp: {ms:8.2f} ms per build = {ms * 1e9 / (n * glen):6.2f} ps per base   {t}")
except Exception as e:
    print(n, glen, "failed:", out[-300:])
PY
done
