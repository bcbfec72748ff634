#!/usr/bin/env python
"""FASTQ path timing (BASELINE.json configs[4] shape): one isolate = 2 x 150 bp reads at `cov`x of a 5 Mbp genome,
0.5 % substitution errors, per-cycle Phred profile; k = 41 (u128), --min-count 5, --qual-filter strict --min-qual 20.
usage: tools/reads_bench.py [n_samples] [coverage] [k]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import skx_engine as E  # noqa: E402
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
k = int(sys.argv[3]) if len(sys.argv) > 3 else 41
E.load_library()
ctx = E.Context(0)
glen, rl = 5_000_000, 150
anc = synth.ancestor(glen, seed=1)
comp = np.zeros(256, np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    comp[a] = b


def sample_reads(i):
    rng = np.random.default_rng([1, 99, i])
    g = synth.sample_bases(anc, i, 1000)
    nreads = int(cov * glen / rl)
    start = rng.integers(0, glen - rl, size=nreads)
    idx = start[:, None] + np.arange(rl)[None, :]
    reads = g[idx]
    rev = rng.random(nreads) < 0.5
    reads[rev] = comp[reads[rev][:, ::-1]]
    err = rng.random(reads.shape) < 0.005
    reads[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(err.sum()))]
    prof = np.clip(38 - (np.arange(rl) // 10), 2, 40)                      # per-cycle Phred, decaying
    q = np.clip(prof[None, :] + rng.integers(-6, 3, size=reads.shape), 2, 41).astype(np.uint8) + 33
    q[err] = 33 + 8
    seq = np.full((nreads, rl + 1), 10, np.uint8); seq[:, :rl] = reads
    qs = np.full((nreads, rl + 1), ord("~"), np.uint8); qs[:, :rl] = q
    return seq.tobytes(), qs.tobytes()


data = [sample_reads(i) for i in range(n)]
q = E.qual(min_count=5, min_qual=20, qual_filter=E.QUAL_STRICT)
for rep in range(2):
    ctx.timings(reset=True)
    t0 = time.perf_counter()
    ds = E.DictSet.build([d[0] for d in data], k, True, q=q, ctx=ctx, quals=[d[1] for d in data])
    ctx.sync()
    dt = time.perf_counter() - t0
    sizes = [ds.size(i) for i in range(n)]
    ds.free()
bases = sum(len(d[0]) for d in data)
print(f"{n} isolates, {bases / n / 1e6:.0f} Mbases each, k={k}: {dt:.2f} s wall incl. H2D ({n / dt:.2f} isolates/s, {bases / dt / 1e9:.2f} Gbases/s); "
      f"split k-mers kept per isolate {sizes}")
