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

The whole body of a rank -- dictionaries, the three exchanges (RCCL, issued by the engine on its own stream), filter, output -- is
behind the C ABI (skh_build_sharded / skh_align_sharded / skh_distance_sharded over skx_comm_*; the `ska` executable reaches the
same code with `--gpus N`).  torch.distributed is the launcher's rendezvous: it carries the 128-byte RCCL id from rank 0 to the
others.  SKX_MULTI_BACKEND=gloo + SKX_MULTI_DEVICE=d let several ranks share one GPU (tests): the engine then exchanges through
its host-staged transport, since RCCL refuses two ranks on one device.
"""
import argparse
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
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = int(os.environ.get("SKX_MULTI_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("SKX_MULTI_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("ska_multi.py needs gfx950 GPUs: the engine has no CPU path")
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # torch.distributed is the rendezvous only (it carries the 128-byte RCCL id to the ranks); the exchanges are the engine's
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import dist as skdist
    import skx_engine as E
    E.load_library()
    ctx = E.Context(device)
    comm = skdist.make_comm(ctx, rank, world, transport="rccl" if backend == "nccl" else "local")

    names, f1, f2 = read_inputs(args, E)
    n_total = len(names)
    lo, hi = skdist.shard_range(n_total, rank, world)
    if hi <= lo:
        raise SystemExit(f"rank {rank}: no samples (fewer samples than ranks)")
    rep = {"world": world, "samples": n_total, "rank0_samples": hi - lo, "transport": "rccl" if backend == "nccl" else "local"}
    t0 = time.perf_counter()
    q = E.qual(min_count=args.min_count, min_qual=args.min_qual, qual_filter=QUALS[args.qual_filter])
    common = dict(k=args.k, rc=not args.single_strand, q=q, threads=args.threads)
    inputs = list(zip(f1, f2))
    E.phases(reset=True)
    if args.command == "build":
        comm.build(names, inputs, args.output, merge_parts=args.merge, **common)
    elif args.command == "align":
        comm.align(names, inputs, args.output, min_freq=0.9 if args.min_freq is None else args.min_freq, filter_type=FILTERS[args.filter],
                   mask_ambig=args.ambig_mask, ignore_const_gaps=args.no_gap_only_sites, filter_ambig_as_missing=args.filter_ambig_as_missing, **common)
    else:
        comm.distance(names, inputs, args.output, min_freq=0.0 if args.min_freq is None else args.min_freq, filt_ambig=not args.allow_ambiguous, **common)
    rep["total_s"] = time.perf_counter() - t0
    rep["bytes_received_rank0"] = comm.bytes_received
    rep["phases_rank0"] = E.phases()
    if rank == 0 and args.report:
        import json
        json.dump(rep, open(args.report, "w"))
    comm.free()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
