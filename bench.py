#!/usr/bin/env python
"""bench.py -- `ska build` + `ska align` throughput of the MI355X engine (BASELINE.json metric).

One *step* = one pass of the hot path over one batch of synthetic assemblies that are already resident in HBM:
  split-k-mer extraction + per-sample dictionaries  ->  key union + samples x k-mers matrix (MergeSkaArray)
  ->  variant-site filter (min_freq 0.9, no-const) + alignment compaction.
Workload at N=1: BASELINE.json configs[2] (1 000 related 5 Mbp assemblies, k=31).  With --gpus N each rank owns
the same number of samples (weak scaling); the only collective is one all-gather of the per-rank key tables plus
the reduction of the per-row filter statistics (ska.rust_amd/dist.py).

Prints ONE JSON line (rank 0).  `roofline` is for the split-k-mer extraction+scatter kernel (HBM bound):
algorithmic bytes = 10 B per input base (1 B ASCII read + 9 B (key, middle base) written) over the
kernel's launch duration measured with HIP events on the engine's stream.  `cpu_baseline` times the CPU oracle
(a restatement of ska.rust's algorithm, NOT the Rust binary) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1000, help="samples per GPU")
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--cpu-genomes", type=int, default=64, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--check", action="store_true", help="verify a subsample of the result against the CPU oracle")
    return ap.parse_args()


def private_snps(n_total):
    """500 private SNPs per sample up to 1 000 samples (configs 2-3), 100 beyond (config 4), so that the
    rows x samples matrix of the sharded runs stays within one GPU's HBM."""
    return 500 if n_total <= 1000 else 100


def other_kernels(tm, steps, n_bases, n_distinct, rows, rows_kept, n_samples):
    """HBM rate of the remaining stages against SURVEY.md 8d's algorithmic bytes (W + 1 = 9 B per dictionary entry,
    P ~ N windows, D = sum of distinct split k-mers per sample, U rows, U' rows kept)."""
    out = []
    for name, ms, nbytes in (
            ("per-sample dedup (dedupe_mb_kernel)", tm["dedupe"] / steps, 9.0 * (n_bases + n_distinct)),
            ("merge (union_kernel + assemble_kernel)", (tm["key_union"] + tm["assemble"]) / steps, 9.0 * n_distinct + rows * (8.0 + n_samples)),
            ("filter + compaction", (tm["filter"] + tm["compact"]) / steps, rows * (8.0 + n_samples) + rows_kept * float(n_samples))):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out.append({"stage": name, "ms": ms, "algorithmic_bytes": nbytes, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
    return out


def cpu_baseline(args, anc, n_total):
    """Oracle build_and_merge + align on a bounded sample of the same workload, timed on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ora
    import synth
    n = args.cpu_genomes
    priv = private_snps(n_total)
    cores = os.cpu_count() or 1
    want = max(1, min(cores, 1 + n // 10))
    threads = 1 << int(np.floor(np.log2(want)))          # merge_ska_dict.rs:384-385
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        inputs = []
        for i in range(n):
            p = os.path.join(td, f"g{i}.fa")
            synth.to_fasta(synth.sample_stream(anc, i, n_total, private_snps=priv), p)
            inputs.append((f"g{i}", p, None))
        t0 = time.perf_counter()
        arr = ora.Array.build(inputs, k=args.k, rc=True, threads=threads)
        t1 = time.perf_counter()
        aln = arr.align(min_freq=0.9)
        t2 = time.perf_counter()
    return {"value": n / (t2 - t0), "unit": "genomes/s", "cores": threads, "kind": "port",
            "sample": f"{n} of the {n_total} synthetic {args.genome_len} bp assemblies, build {t1 - t0:.2f}s + align {t2 - t1:.2f}s, "
                      f"{threads} thread(s) per the reference's 10-samples-per-thread rule, alignment {len(aln)} B"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("SKX_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # override: ranks sharing one GPU in tests
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or os.environ.get("SKX_BENCH_FORCE_SHARDED") == "1"     # the override runs the exchange path at world size 1
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SKX_BENCH_BACKEND", "nccl")      # "gloo" lets two ranks share one GPU in tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import dist as skdist
    import skx_engine as E
    import synth
    E.load_library()
    ctx = E.Context(local_rank)

    G = args.genomes
    n_total = G * world
    lo = rank * G
    anc = synth.ancestor(args.genome_len, seed=1)
    # ---- inputs -> HBM (not timed): one 16-B aligned record stream per sample
    lens, offs, tot = [], [], 0
    streams = []
    for i in range(G):
        s = synth.sample_stream(anc, lo + i, n_total, private_snps=private_snps(n_total))
        streams.append(s)
        offs.append(tot)
        lens.append(len(s))
        tot += (len(s) + 255) // 256 * 256
    pool = torch.empty(tot + 256, dtype=torch.uint8, device=dev)
    for i, s in enumerate(streams):
        pool[offs[i]:offs[i] + lens[i]] = torch.from_numpy(s).to(dev, non_blocking=False)
    del streams
    base = pool.data_ptr()
    assert base % 256 == 0
    ptrs = [base + o for o in offs]
    names = [f"g{lo + i}" for i in range(G)]
    total_bases = int(sum(lens))
    torch.cuda.synchronize()

    host_ms = {"build": 0.0, "merge": 0.0, "align": 0.0, "free": 0.0}

    def step():
        t_a = time.perf_counter()
        ds = E.DictSet.build_device(ptrs, lens, args.k, True, ctx=ctx)
        ctx.sync()
        t_b = time.perf_counter()
        host_ms["build"] += (t_b - t_a) * 1e3
        if not sharded:
            arr = ds.merge(names)
        else:
            ks = ds.union_keys()
            p, n, _ = ks.device()
            ctx.sync()
            tables = skdist.allgather_tables(skdist.as_tensor(p, n, "<i8", dev))
            torch.cuda.synchronize()
            sets = [E.KeySet.from_device(t.data_ptr(), t.numel(), args.k, True, ctx=ctx) for t in tables]
            rows = E.KeySet.merge(sets, ctx=ctx)
            arr = ds.assemble(rows, names)
            pp, pu, pm, pv = arr.device_stats()
            U = arr.nrows
            tp, tu, tm = (skdist.as_tensor(x, U, "<i4", dev) for x in (pp, pu, pm))
            skdist.reduce_row_stats(tp, tu, tm, total_samples=n_total)
            skdist.as_tensor(pv, U, "<i4", dev).copy_(tp)
            torch.cuda.synchronize()
            arr.set_total_samples(n_total)
        ctx.sync()
        t_c = time.perf_counter()
        host_ms["merge"] += (t_c - t_b) * 1e3
        info = (arr.nrows, arr.nsamples)
        removed = arr.apply_filters(0.9, False, E.FILTER_NO_CONST, False, False)      # ska align defaults
        ctx.sync()
        t_d = time.perf_counter()
        host_ms["align"] += (t_d - t_c) * 1e3
        out = (info[0], arr.nrows, removed)
        ds.free()
        host_ms["free"] += (time.perf_counter() - t_d) * 1e3
        return arr, out

    for _ in range(args.warmup):
        arr, _ = step()
        arr.free()
    ctx.timings(reset=True)
    for kk in host_ms:
        host_ms[kk] = 0.0
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        if last is not None:
            last.free()
        last, shape = step()
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    check = None
    if args.check and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ora
        nchk = min(G, 3)
        od = []
        for i in range(nchk):
            d = ora.Dict.new(args.k, True)
            for rec in synth.sample_stream(anc, lo + i, n_total, private_snps=private_snps(n_total)).tobytes().split(b"\n")[:-1]:
                d.add_record(rec)
            od.append(d)
        ds = E.DictSet.build_device(ptrs[:nchk], lens[:nchk], args.k, True, ctx=ctx)
        ok = True
        for i in range(nchk):
            gk, gb = ds.export(i)
            okk, ob = od[i].export()
            ok &= bool(np.array_equal(gk["lo"], okk["lo"]) and np.array_equal(gb, ob))
        check = {"dicts_equal_oracle": ok, "samples_checked": nchk}

    n_distinct = None
    if rank == 0:                      # sum of the per-sample dictionary sizes (untimed): the D of SURVEY.md 8d's per-stage bytes
        ds = E.DictSet.build_device(ptrs, lens, args.k, True, ctx=ctx)
        n_distinct = int(sum(ds.size(i) for i in range(G)))
        ds.free()
    res = None
    if rank == 0:
        steps = max(args.steps, 1)
        scatter_ms = tm["scatter"] / steps
        algo_bytes = 10.0 * total_bases
        achieved = algo_bytes / (scatter_ms * 1e-3) / 1e9 if scatter_ms > 0 else 0.0
        traffic = None       # HBM bytes per launch from the PMC passes recorded in profiles/ (FETCH_SIZE x2 + WRITE_SIZE), scaled per base
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_extract.json")))
            traffic = (pmc["read_bytes_per_base"] + pmc["write_bytes_per_base"]) * total_bases
        except Exception:
            pass
        res = {
            "metric": "genomes/sec ska build+align, 1 000x5 Mbp k=31; bit-exact vs CPU",
            "value": n_total * steps / dt, "unit": "genomes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"ska build + ska align, {G} synthetic {args.genome_len} bp assemblies per GPU, k={args.k}, "
                                   f"inputs resident in HBM (BASELINE.json configs[2])",
                       "samples_per_gpu": G, "private_snps": private_snps(n_total), "genome_len": args.genome_len, "k": args.k,
                       "rows_U": shape[0], "rows_kept": shape[1], "parallelism": f"samples sharded x{world}" + (" (key-table all-gather path)" if sharded else "")},
            "roofline": {"bound": "hbm", "kernel": "extract_kernel<true> (split k-mer extraction + bucket scatter)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": scatter_ms},
            "stage_ms_per_step": {k: v / steps for k, v in tm.items()},
            "other_kernels": other_kernels(tm, steps, total_bases, n_distinct, shape[0], shape[1], G),
            "host_wall_ms_per_step": {k: v / steps for k, v in host_ms.items()},
        }
        if check:
            res["check"] = check
        if world == 1 and args.cpu_genomes > 0:
            res["cpu_baseline"] = cpu_baseline(args, anc, n_total)
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is flushed at exit: push it out first so that the
        # JSON line is the last line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
