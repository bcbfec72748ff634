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
    if total_samples is not None and 0 < total_samples <= 0x7FFF:
        # both sums stay below 2^15, so the packed int32 never reaches its sign bit (no reliance on wrap-around)
        packed = present + (unambig << 16)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        present.copy_(packed & 0xFFFF)
        unambig.copy_((packed >> 16) & 0xFFFF)
    else:
        dist.all_reduce(present, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(unambig, op=dist.ReduceOp.SUM, group=group)
    # 16 code bits per row, sent as two explicit bytes (neither backend moves int16; no narrowing cast of bit 15)
    m8 = torch.stack(((mask & 0xFF).to(torch.uint8), ((mask >> 8) & 0xFF).to(torch.uint8)), dim=1).reshape(-1)
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
    acc = acc.view(-1, 2).to(torch.int32)
    mask.copy_(acc[:, 0] | (acc[:, 1] << 8))
    return present, unambig, mask


# ---- all-vs-all distance over ranks ("tile the pair matrix over ranks") ---------------------------------------
def pair_bands(n_samples, world, align=8):
    """Rows of the pair matrix dealt to ranks: contiguous bands [i_lo, i_hi), starts on multiples of `align`, about the same
    number of pairs (i, j > i) each (row i holds n - 1 - i of them)."""
    total = n_samples * (n_samples - 1) // 2

    def before(h):                      # pairs in rows [0, h)
        return h * (n_samples - 1) - h * (h - 1) // 2

    bands, lo = [], 0
    for r in range(world):
        if r == world - 1:
            hi = n_samples
        else:
            want = total * (r + 1) // world
            h = lo
            while h < n_samples and before(h) < want:
                h += 1
            down, up = h // align * align, min(n_samples, (h + align - 1) // align * align)
            hi = down if down >= lo and abs(before(down) - want) <= abs(before(up) - want) else up
        hi = max(hi, lo)
        bands.append((lo, hi))
        lo = hi
    return bands


def allgather_planes(local: torch.Tensor, group=None):
    """local: [n_planes, n_local_samples, words] int64 bit planes of this rank's samples -> [n_planes, n_total_samples, words],
    samples in rank order (ranks may hold different numbers of samples: padded to the largest for the collective)."""
    world = dist.get_world_size(group)
    P, s_loc, W = local.shape
    n = torch.tensor([s_loc], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    mx = max(max(sizes), 1)
    padded = torch.zeros((P, mx, W), dtype=local.dtype, device=local.device)
    padded[:, :s_loc] = local
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
    else:
        out = torch.empty((world, P, mx, W), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(-1), padded.view(-1), group=group)      # one RCCL all-gather
        parts = [out[r] for r in range(world)]
    return torch.cat([parts[r][:, : sizes[r]] for r in range(world)], dim=1).contiguous(), sizes


def distance_sharded(local_planes, pair_fn, group=None):
    """All-vs-all distances of a job whose samples are sharded over ranks.  local_planes: this rank's bit planes over the globally
    filtered rows; pair_fn(planes [P, S, W], i_lo, i_hi) -> float64 array [n_pairs, 4] (distance, mismatch proportion, matches,
    mismatches) of the pairs (i in [i_lo, i_hi), j > i), row-major -- the engine's skx_planes_distance on a GPU, numpy in the
    gloo test.  Returns the whole table [S (S - 1) / 2, 4] on rank 0 (pairs in the reference's (i < j) row-major order), None elsewhere."""
    import numpy as np
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    planes, sizes = allgather_planes(local_planes, group)
    S = int(sum(sizes))
    bands = pair_bands(S, world)
    lo, hi = bands[rank]
    mine = np.ascontiguousarray(pair_fn(planes, lo, hi), dtype=np.float64).reshape(-1, 4)
    counts = [sum(S - 1 - i for i in range(a, b)) for a, b in bands]
    assert mine.shape[0] == counts[rank]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, 4), dtype=torch.float64)
    buf[: mine.shape[0]] = torch.from_numpy(mine)
    dev = local_planes.device if dist.get_backend(group) != "gloo" else torch.device("cpu")
    buf = buf.to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)                # 32 bytes per pair: 16 MB for 1 000 samples
    if rank != 0:
        return None
    return torch.cat([parts[r][: counts[r]].cpu() for r in range(world)], dim=0).numpy()


def distance_tsv(names, table):
    """generic_modes::distance's long-form table (generic_modes.rs:170-188; VariantDist Display "{:.2}\t{:.5}\t{}\t{}")"""
    out = ["Sample1\tSample2\tDistance\tMismatches (proportion)\tMatch count\tMismatch count\n"]
    n = 0
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            d, p, m, mm = table[n]
            out.append("%s\t%s\t%.2f\t%.5f\t%d\t%d\n" % (names[i], names[j], d, p, int(m), int(mm)))
            n += 1
    return "".join(out).encode()


class DevicePtr:
    """Zero-copy view of engine-owned device memory as a torch tensor (via __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def as_tensor(ptr, n, typestr, device):
    if n == 0:
        return torch.empty(0, dtype={"<i8": torch.int64, "<i4": torch.int32, "|u1": torch.uint8}[typestr], device=device)
    return torch.as_tensor(DevicePtr(ptr, n, typestr), device=device)
