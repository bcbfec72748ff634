#!/usr/bin/env python
"""`ska map` at scale: n synthetic genomes mapped onto their ancestor. usage: tools/map_bench.py [n] [genome_len] [aln|vcf]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import skx_engine as E  # noqa: E402
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
fmt = sys.argv[3] if len(sys.argv) > 3 else "aln"
E.load_library()
ctx = E.Context(0)
anc = synth.ancestor(glen, seed=1)
ds = E.DictSet.build([synth.sample_stream(anc, i, n).tobytes() for i in range(n)], 31, True, ctx=ctx)
arr = ds.merge([f"g{i}" for i in range(n)])
ds.free()
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    ref = os.path.join(td, "ref.fa")
    synth.to_fasta(synth.sample_stream(anc, 0, 1, private_snps=0, shared_snps=0, decorate=False), ref)
    for rep in range(2):
        t0 = time.perf_counter()
        out = arr.map(ref, fmt=fmt)
        dt = time.perf_counter() - t0
    print(f"ska map ({fmt}): {n} samples x {glen} bp reference, rows {arr.nrows}: {dt:.2f} s, {len(out) / 1e6:.1f} MB of text")
