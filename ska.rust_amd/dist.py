"""Multi-GPU plumbing of the build -> merge -> align path.

Samples shard contiguously by rank (independent units, no collective on the data path); the only exchange is
one all-gather of the per-rank key tables (RCCL over xGMI when the tensors live on GPUs, gloo on CPU in tests)
so that every rank derives the same global row set, plus a small reduction of the per-row statistics the
variant-site filter needs.  Pure torch.distributed on plain integer tensors: the same code runs on CPU tensors
under gloo (tests/test_dist_cpu.py) and on device tensors under nccl (bench.py --gpus N).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank; preserves input order so that names stay in CLI order
    (cf. the offset handling of merge_ska_dict.rs:243-253,277-291)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_tables(local: torch.Tensor, group=None):
    """All-gather variable-length 1-D int64 tables (padded to the longest).  Returns a list of per-rank tensors."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros(mx, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    out = torch.empty(world * mx, dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
    else:
        dist.all_gather_into_tensor(out, padded, group=group)     # one RCCL all-gather
        parts = [out[r * mx:(r + 1) * mx] for r in range(world)]
    return [parts[r][: sizes[r]] for r in range(world)]


def reduce_row_stats(present: torch.Tensor, unambig: torch.Tensor, mask: torch.Tensor, group=None, total_samples=None):
    """Global per-row statistics from per-rank column slabs: counts add, code masks OR (in place).
    NCCL has no bitwise reduction, so the 16-bit code sets are all-gathered (two bytes per row) and OR-ed locally.  When the whole
    job has fewer than 65 536 samples (`total_samples`), both counts travel in one all-reduce, 16 bits each."""
    world = dist.get_world_size(group)
    if total_samples is not None and 0 < total_samples <= 0xFFFF:
        packed = present + (unambig << 16)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        present.copy_(packed & 0xFFFF)
        unambig.copy_((packed >> 16) & 0xFFFF)
    else:
        dist.all_reduce(present, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(unambig, op=dist.ReduceOp.SUM, group=group)
    m8 = mask.to(torch.int16).view(torch.uint8)             # 16 code bits per row, sent as bytes (neither backend moves int16)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(m8) for _ in range(world)]
        dist.all_gather(parts, m8, group=group)
    else:
        out = torch.empty(world * m8.numel(), dtype=torch.uint8, device=m8.device)
        dist.all_gather_into_tensor(out, m8, group=group)
        parts = [out[r * m8.numel():(r + 1) * m8.numel()] for r in range(world)]
    acc = parts[0].clone()
    for p in parts[1:]:
        acc |= p
    acc = acc.view(torch.int16)
    mask.copy_(acc.to(torch.int32) & 0xFFFF)
    return present, unambig, mask


class DevicePtr:
    """Zero-copy view of engine-owned device memory as a torch tensor (via __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def as_tensor(ptr, n, typestr, device):
    if n == 0:
        return torch.empty(0, dtype={"<i8": torch.int64, "<i4": torch.int32}[typestr], device=device)
    return torch.as_tensor(DevicePtr(ptr, n, typestr), device=device)
