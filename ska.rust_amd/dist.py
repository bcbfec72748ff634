"""Multi-GPU plumbing of the build -> merge -> align / distance path for Python hosts.

The exchanges themselves live behind the C ABI (include/skx.h "Collectives", csrc/skx_comm.hip: RCCL over xGMI on the engine's
stream, or a host-staged transport for ranks that share one device); this module only does what a host has to do around them:
carry the 128-byte RCCL id from rank 0 to the other ranks (through the torch.distributed store the launcher already set up) and
format the distance table.  torch is plumbing here: rendezvous, barrier and the max-over-ranks of bench.py's timing.
"""
import os

import skx_engine as E


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank; preserves input order so that names stay in CLI order
    (cf. the offset handling of merge_ska_dict.rs:243-253,277-291).  skx_shard_range."""
    return E.shard_range(n_items, rank, world)


def pair_bands(n_samples, world, align=8):
    """Rows of the pair matrix dealt to ranks (skx_pair_bands): contiguous bands [i_lo, i_hi), starts on multiples of `align`, about
    the same number of pairs (i, j > i) each."""
    return E.pair_bands(n_samples, world, align)


def make_comm(ctx, rank=None, world=None, transport=None, directory=None):
    """The engine's communicator for this rank of a torch.distributed job.

    transport "rccl" (default): rank 0 calls skx_comm_unique_id, the id travels through torch.distributed (any backend: it is 128
    bytes of host data), every rank calls skx_comm_create on its own device.  transport "local": host-staged exchange through
    `directory` (ranks sharing one GPU: RCCL refuses that); rank 0's choice of directory is broadcast the same way."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    transport = transport or os.environ.get("SKX_COMM", "rccl")
    if transport == "local":
        box = [directory]
        if box[0] is None and rank == 0:
            import tempfile
            box[0] = tempfile.mkdtemp(prefix="skx_comm_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return E.Comm.local(rank, world, box[0], ctx=ctx)
    box = [E.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return E.Comm.rccl(rank, world, box[0], ctx=ctx)


def distance_tsv(names, table):
    """generic_modes::distance's long-form table (generic_modes.rs:170-188; VariantDist Display "{:.2}\\t{:.5}\\t{}\\t{}");
    table: the DIST_DT records skx_array_distance_sharded leaves on rank 0"""
    out = ["Sample1\tSample2\tDistance\tMismatches (proportion)\tMatch count\tMismatch count\n"]
    n = 0
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            t = table[n]
            out.append("%s\t%s\t%.2f\t%.5f\t%d\t%d\n" % (names[i], names[j], t["distance"], t["mismatch_prop"], int(t["match_count"]), int(t["mismatch_count"])))
            n += 1
    return "".join(out).encode()


class DevicePtr:
    """Zero-copy view of engine-owned device memory as a torch tensor (via __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def as_tensor(ptr, n, typestr, device):
    import torch
    if n == 0:
        return torch.empty(0, dtype={"<i8": torch.int64, "<i4": torch.int32, "|u1": torch.uint8}[typestr], device=device)
    return torch.as_tensor(DevicePtr(ptr, n, typestr), device=device)
