#!/usr/bin/env python
"""ska_multi.py -- the build -> merge -> align / distance path on several GPUs of one node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        ska.rust_amd/ska_multi.py build    -f file_list.txt -o out -k 31 [--threads T] [--min-count C ...]
        ska.rust_amd/ska_multi.py align    -f file_list.txt -o aln.fa [-k 31] [--min-freq 0.9] [--filter no-const] ...
        ska.rust_amd/ska_multi.py distance -f file_list.txt -o dists.tsv [-k 31] [--min-freq 0] [--allow-ambiguous]

Samples are dealt to the ranks in contiguous shards (input order is kept, so names come out in CLI order, cf. the offset
handling of merge_ska_dict.rs:243-253,277-291).  Nothing on the data path is collective; the exchanges are:
  * one all-gather of the per-rank key tables -> every rank derives the same global row set and fills its own column slab;
  * the per-row filter statistics (two counts in one all-reduce, 16-bit code sets all-gathered and OR-ed);
  * `distance`: one all-gather of the per-rank bit planes over the filtered rows; the pair matrix is tiled over ranks by bands
    of first samples (merge_ska_array.rs:416-438 order kept) and the finished pairs are gathered on rank 0.
No rank ever holds its column slab of the global rows in full (8 000 samples: 100+ GB per rank): the array stays rows + dictionaries,
the row statistics are gathered on their own, the filter then writes the kept rows only, `build` streams windows into its file.
`align` needs no gather at all: the alignment is sample-major, every rank writes its own samples' records at their offsets of
the one output file.  `build` leaves one .skf per rank (`<out>.part<r>of<N>.skf`: the global rows x that rank's samples; each is
a valid MergeSkaArray, and `ska merge` -- this engine's or the reference's -- joins them); with --merge rank 0 joins them itself
when the whole matrix fits one GPU.

The engine is reached through the C ABI (skx_engine.py -> libskx.so); torch is here for torch.distributed ("nccl" is RCCL on ROCm)
and device tensors only.  SKX_MULTI_BACKEND=gloo + SKX_MULTI_DEVICE=d let several ranks share one GPU (tests).
"""
import argparse
import math
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

FILTERS = {"no-filter": 0, "no-const": 1, "no-ambig": 2, "no-ambig-or-const": 3}
QUALS = {"no-filter": 0, "middle": 1, "strict": 2}


def parse():
    ap = argparse.ArgumentParser(prog="ska_multi.py")
    ap.add_argument("command", choices=["build", "align", "distance"])
    ap.add_argument("seq_files", nargs="*")
    ap.add_argument("-f", dest="file_list")
    ap.add_argument("-o", dest="output", required=True)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--single-strand", action="store_true")
    ap.add_argument("--min-count", type=int, default=5)
    ap.add_argument("--min-qual", type=int, default=20)
    ap.add_argument("--qual-filter", choices=list(QUALS), default="strict")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("-m", "--min-freq", type=float, default=None)
    ap.add_argument("--filter", choices=list(FILTERS), default="no-const")
    ap.add_argument("--filter-ambig-as-missing", action="store_true")
    ap.add_argument("--ambig-mask", action="store_true")
    ap.add_argument("--no-gap-only-sites", action="store_true")
    ap.add_argument("--allow-ambiguous", action="store_true")
    ap.add_argument("--merge", action="store_true", help="build: rank 0 joins the per-rank files into <out>.skf")
    ap.add_argument("--report", help="rank 0 writes exchange sizes / times as JSON here")
    return ap.parse_args()


def read_inputs(args, E):
    names, f1, f2 = [], [], []
    if args.file_list:                                              # io_utils.rs:116-146
        for line in open(args.file_list):
            fld = line.split()
            if not fld:
                continue
            if len(fld) not in (2, 3):
                raise SystemExit("Unable to parse line in file_list")
            names.append(fld[0]); f1.append(fld[1]); f2.append(fld[2] if len(fld) == 3 else None)
    else:
        for p in args.seq_files:
            names.append(E.sample_name(p)); f1.append(p); f2.append(None)
    if not names:
        raise SystemExit("give either sequence files or -f <file_list>")
    return names, f1, f2


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = int(os.environ.get("SKX_MULTI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("SKX_MULTI_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("ska_multi.py needs gfx950 GPUs: the engine has no CPU path")
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import dist as skdist
    import skx_engine as E
    E.load_library()
    ctx = E.Context(device)
    xdev = dev if backend == "nccl" else torch.device("cpu")        # where the collectives' tensors live

    names, f1, f2 = read_inputs(args, E)
    n_total = len(names)
    lo, hi = skdist.shard_range(n_total, rank, world)
    if hi <= lo:
        raise SystemExit(f"rank {rank}: no samples (fewer samples than ranks)")
    rep = {"world": world, "samples": n_total, "rank0_samples": hi - lo}
    t0 = time.perf_counter()
    q = E.qual(min_count=args.min_count, min_qual=args.min_qual, qual_filter=QUALS[args.qual_filter])
    ds = E.DictSet.from_files(list(zip(f1[lo:hi], f2[lo:hi])), args.k, not args.single_strand, q=q, threads=args.threads, ctx=ctx)
    rep["build_dictionaries_s"] = time.perf_counter() - t0
    # ---- exchange 1: key tables -> global rows
    t1 = time.perf_counter()
    ks = ds.union_keys()
    p, n_keys, wpk = ks.device()
    ctx.sync()
    local = skdist.as_tensor(p, n_keys * wpk, "<i8", dev).to(xdev)
    tables = skdist.allgather_tables(local)
    torch.cuda.synchronize()
    rep["key_table_allgather_bytes_per_rank"] = int(sum(t.numel() for t in tables) * 8)
    rep["key_table_allgather_s"] = time.perf_counter() - t1
    tabs = [t.to(dev).contiguous() for t in tables]
    sets = [E.KeySet.from_device(t.data_ptr(), t.numel() // wpk, args.k, not args.single_strand, ctx=ctx) for t in tabs]
    rows = E.KeySet.merge(sets, ctx=ctx)
    arr = ds.assemble_lazy(rows, names[lo:hi])          # rows + dictionaries: the rank's column slab over the global rows is never allocated
    U = arr.nrows
    rep["rows"] = int(U)

    if args.command == "build":
        part = f"{args.output}.part{rank}of{world}.skf"
        arr.save(part)                                              # local counts: each part is a self-consistent MergeSkaArray
        dist.barrier()
        if rank == 0:
            parts = [f"{args.output}.part{r}of{world}.skf" for r in range(world)]
            if args.merge:
                out = args.output if args.output.endswith(".skf") else args.output + ".skf"       # generic_modes.rs:272-276
                E.Array.merge([E.Array.load(x, ctx=ctx) for x in parts], ctx=ctx).save(out)
                for x in parts:
                    os.unlink(x)
                print(f"wrote {out}", file=sys.stderr)
            else:
                print("wrote " + " ".join(parts) + f"\njoin them with: ska merge -o {args.output} " + " ".join(parts), file=sys.stderr)
    else:
        # ---- exchange 2: per-row filter statistics over all ranks
        t2 = time.perf_counter()
        pp, pu, pm, pv = arr.device_stats()
        tp, tu, tm = (skdist.as_tensor(x, U, "<i4", dev) for x in (pp, pu, pm))
        cp, cu, cm = (t.to(xdev) for t in (tp, tu, tm))
        skdist.reduce_row_stats(cp, cu, cm, total_samples=n_total)
        tp.copy_(cp); tu.copy_(cu); tm.copy_(cm)
        skdist.as_tensor(pv, U, "<i4", dev).copy_(tp)
        torch.cuda.synchronize()
        arr.set_total_samples(n_total)
        rep["row_stats_bytes_per_rank"] = int(U * (4 + 2 * world))
        rep["row_stats_s"] = time.perf_counter() - t2

        if args.command == "align":
            mf = 0.9 if args.min_freq is None else args.min_freq
            arr.apply_filters(mf, args.filter_ambig_as_missing, FILTERS[args.filter], args.ambig_mask, args.no_gap_only_sites)
            kept = arr.nrows
            # every rank writes its samples' records at their place in the one file (merge_ska_array.rs:499-517 order)
            sizes = [len(nm.encode()) + kept + 3 for nm in names]
            offset = sum(sizes[:lo])
            if rank == 0:
                with open(args.output, "wb") as f:
                    f.truncate(sum(sizes))
            dist.barrier()
            fd = os.open(args.output, os.O_RDWR)
            try:
                os.lseek(fd, offset, os.SEEK_SET)
                arr.write_fasta(fd)
            finally:
                os.close(fd)
            dist.barrier()
            rep["alignment_columns"] = int(kept)
        else:
            mf = 0.0 if args.min_freq is None else args.min_freq
            filt = not args.allow_ambiguous
            if mf * n_total >= 1.0:                                     # generic_modes.rs:149-159
                arr.filter(math.ceil(n_total * mf), False, E.FILTER_NONE, False, False, False)
            constant = arr.filter(0, False, E.FILTER_NO_CONST, False, False, False)       # :161-168
            t3 = time.perf_counter()
            pl, wpr, n_planes = arr.distance_planes(filt)
            local = skdist.as_tensor(pl, n_planes * (hi - lo) * wpr, "<i8", dev).view(n_planes, hi - lo, wpr).to(xdev)
            keep = {}

            def pair_fn(planes, i_lo, i_hi):
                g = planes.to(dev).contiguous()
                keep["planes"] = g
                d = E.planes_distance(g.data_ptr(), g.shape[1], g.shape[2], filt, constant, i_lo, i_hi, ctx=ctx)
                return np.stack([d["distance"], d["mismatch_prop"], d["match_count"].astype(np.float64), d["mismatch_count"].astype(np.float64)], axis=1)

            table = skdist.distance_sharded(local, pair_fn)
            rep["planes_allgather_bytes_per_rank"] = int(n_planes * n_total * wpr * 8)
            rep["distance_s"] = time.perf_counter() - t3
            if rank == 0:
                with open(args.output, "wb") as f:
                    f.write(skdist.distance_tsv(names, table))
    rep["total_s"] = time.perf_counter() - t0
    if rank == 0 and args.report:
        import json
        json.dump(rep, open(args.report, "w"))
    arr.free()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
