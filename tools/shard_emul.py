#!/usr/bin/env python
"""What one rank of an N-GPU `bench.py --gpus N` run does, emulated on one GPU: the other ranks' key tables are built
here one after another (their dictionaries are dropped again), then rank 0 merges the N tables, assembles its own column
slab over the global row set and filters.  Checks memory and time of the sharded path at config-4 scale without N GPUs.
usage: tools/shard_emul.py [n_ranks] [samples_per_rank]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dist as skdist  # noqa: E402
import skx_engine as E  # noqa: E402
import synth  # noqa: E402
from bench import private_snps  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_total = world * G
E.load_library()
ctx = E.Context(0)
dev = torch.device("cuda", 0)
anc = synth.ancestor(5_000_000, seed=1)
priv = private_snps(n_total)
tables, ds0 = [], None
for r in range(world):
    t0 = time.perf_counter()
    streams = [synth.sample_stream(anc, r * G + i, n_total, private_snps=priv).tobytes() for i in range(G)]
    t1 = time.perf_counter()
    ds = E.DictSet.build(streams, 31, True, ctx=ctx)
    del streams
    ks = ds.union_keys()
    p, n, _ = ks.device()
    ctx.sync()
    tables.append(skdist.as_tensor(p, n, "<i8", dev).clone())
    t2 = time.perf_counter()
    print(f"rank {r}: synth {t1 - t0:.1f} s, build+union {t2 - t1:.2f} s, local rows {n}", flush=True)
    if r == 0:
        ds0 = ds
    else:
        ds.free()
    ks.free()
torch.cuda.synchronize()
ctx.timings(reset=True)
t0 = time.perf_counter()
sets = [E.KeySet.from_device(t.data_ptr(), t.numel(), 31, True, ctx=ctx) for t in tables]
rows = E.KeySet.merge(sets, ctx=ctx)
ctx.sync()
t1 = time.perf_counter()
arr = ds0.assemble(rows, [f"g{i}" for i in range(G)])
arr.set_total_samples(n_total)
ctx.sync()
t2 = time.perf_counter()
U = arr.nrows
removed = arr.apply_filters(0.9, False, E.FILTER_NO_CONST, False, False)      # (row statistics of this slab only: timing, not the result)
ctx.sync()
t3 = time.perf_counter()
free, total = torch.cuda.mem_get_info()
print(f"{world} ranks x {G} samples ({priv} private SNPs): global rows {U}; merge of {world} tables {1e3 * (t1 - t0):.1f} ms, assemble {1e3 * (t2 - t1):.1f} ms, "
      f"filter+compaction {1e3 * (t3 - t2):.1f} ms; device memory in use {(total - free) / 2**30:.1f} GiB; stage ms {ctx.timings()}")
