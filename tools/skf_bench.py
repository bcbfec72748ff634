#!/usr/bin/env python
""".skf save / load timing of a built array: tools/skf_bench.py [n_genomes] [genome_len]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import skx_engine as E  # noqa: E402
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
E.load_library()
ctx = E.Context(0)
anc = synth.ancestor(glen, seed=1)
ds = E.DictSet.build([synth.sample_stream(anc, i, n).tobytes() for i in range(n)], 31, True, ctx=ctx)
arr = ds.merge([f"g{i}" for i in range(n)])
ds.free()
path = "/dev/shm/skf_bench.skf" if os.path.isdir("/dev/shm") else "/tmp/skf_bench.skf"
t0 = time.perf_counter()
arr.save(path)
t1 = time.perf_counter()
size = os.path.getsize(path)
b = E.Array.load(path, ctx=ctx)
t2 = time.perf_counter()
cells = arr.nrows * n
print(f"rows {arr.nrows} samples {n} cells {cells / 1e6:.0f} M  file {size / 1e6:.1f} MB  save {t1 - t0:.2f} s ({cells / 1e6 / (t1 - t0):.0f} Mcell/s)  "
      f"load {t2 - t1:.2f} s ({cells / 1e6 / (t2 - t1):.0f} Mcell/s)  rows back {b.nrows}")
os.remove(path)
