#!/usr/bin/env python
"""Kernel micro-benchmark: dictionary build (extract+scatter, dedupe) on random genomes generated on the GPU.
usage: kbench.py [n_genomes] [genome_len] [reps] [k]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import torch  # noqa: E402

import skx_engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
K = int(sys.argv[4]) if len(sys.argv) > 4 else 31
E.load_library()
ctx = E.Context(0)
dev = torch.device("cuda", 0)
stride = (glen + 1 + 255) // 256 * 256
g = torch.Generator(device=dev)
g.manual_seed(1)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
pool = lut[torch.randint(0, 4, (n * stride,), device=dev, generator=g)]
pool.view(n, stride)[:, glen] = 10
torch.cuda.synchronize()
ptrs = [pool.data_ptr() + i * stride for i in range(n)]
lens = [glen + 1] * n
for r in range(reps + 1):
    if r == 1:
        ctx.timings(reset=True)
    try:
        ds = E.DictSet.build_device(ptrs, lens, K, True, ctx=ctx)
        ds.free()
    except E.EngineError as e:
        print("engine error (expected in debug modes):", str(e)[:80])
t = ctx.timings()
try:                                     # builds with -DSKX_PHASE_PROF=1|2: cycles of wave 0 per phase of extract_kernel | dedupe_mb_kernel, summed over workgroups and launches
    import ctypes
    f = E._lib.skx_debug_phase_prof
    buf = (ctypes.c_ulonglong * 16)()
    f(buf, 1)
    tot = sum(buf) or 1
    print("phases (share of wave-0 cycles between barriers):", [round(x / tot, 3) for x in buf[:10]], "cycles in all:", tot)
except AttributeError:
    pass
print({k: round(v / reps, 3) for k, v in t.items() if v})
