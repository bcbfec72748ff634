#!/usr/bin/env python
"""`ska distance` at scale: build + merge n synthetic genomes, then time the all-vs-all distance (device stages)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import torch  # noqa: E402

import skx_engine as E  # noqa: E402
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
E.load_library()
ctx = E.Context(0)
anc = synth.ancestor(glen, seed=1)
streams = [synth.sample_stream(anc, i, n).tobytes() for i in range(n)]
ds = E.DictSet.build(streams, 31, True, ctx=ctx)
arr = ds.merge([f"g{i}" for i in range(n)])
ds.free()
print("rows", arr.nrows)
ctx.timings(reset=True)
t0 = time.perf_counter()
tsv = arr.distance_tsv()
t1 = time.perf_counter()
print("distance wall %.1f ms, rows after no-const %d, pairs %d" % ((t1 - t0) * 1e3, arr.nrows, n * (n - 1) // 2), ctx.timings())
print(tsv.decode().split("\n")[1])
