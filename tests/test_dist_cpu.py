"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing in ska.rust_amd/dist.py: contiguous sample shards,
the variable-length all-gather of key tables and the reduction of the per-row filter statistics.  The data are
key tables / statistics produced by the CPU oracle (the checker), so the exchange is verified end to end:
union of the gathered per-shard tables == the row set of one merge over all samples, and reduced statistics ==
statistics of the full matrix."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "ska.rust_amd"))


def _samples(n=6, L=4000, seed=3):
    rng = np.random.default_rng(seed)
    anc = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
    out = []
    for i in range(n):
        s = anc.copy()
        pos = rng.integers(0, L, size=25)
        s[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=25)
        out.append(bytes(s[: L - 100 * i].tolist()))
    return out


def _worker(rank, world, port, tmp, packed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dist as skdist
    import ora
    samples = _samples()
    n = len(samples)
    lo, hi = skdist.shard_range(n, rank, world)
    names = [f"s{i}" for i in range(n)]
    dicts = []
    for s in samples[lo:hi]:
        d = ora.Dict.new(15, True)
        d.add_record(s)
        dicts.append(d)
    local = ora.Array.from_dicts(dicts, names[lo:hi])
    lk, lv, _ = local.export()
    tables = skdist.allgather_tables(torch.from_numpy(lk["lo"].astype(np.int64)))
    rows = np.unique(np.concatenate([t.numpy() for t in tables]))
    # this rank's column slab on the global rows
    idx = np.searchsorted(rows, lk["lo"].astype(np.int64))
    slab = np.full((len(rows), hi - lo), ord("-"), dtype=np.uint8)
    slab[idx] = lv
    present = torch.from_numpy((slab != ord("-")).sum(axis=1).astype(np.int32))
    unambig = torch.from_numpy(np.isin(slab, list(b"ACGT")).sum(axis=1).astype(np.int32))
    codes = {c: i for i, c in enumerate(b"-ACMTWYHGRSVKDBN")}
    mask = np.zeros(len(rows), dtype=np.int32)
    for c, i in codes.items():
        if i:
            mask |= ((slab == c).any(axis=1).astype(np.int32) << i)
    mask = torch.from_numpy(mask)
    skdist.reduce_row_stats(present, unambig, mask, total_samples=len(_samples()) if packed else None)
    np.savez(os.path.join(tmp, f"r{rank}.npz"), rows=rows, present=present.numpy(), unambig=unambig.numpy(), mask=mask.numpy(),
             lo=lo, hi=hi, slab=slab)
    dist.destroy_process_group()


@pytest.mark.parametrize("packed", [False, True])          # counts in two all-reduces | both in one (16 bits each)
def test_sharded_exchange_matches_single_merge(tmp_path, packed):
    import ora
    world = 2
    port = 29500 + (os.getpid() % 1000) + (500 if packed else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), packed), nprocs=world, join=True)
    samples = _samples()
    names = [f"s{i}" for i in range(len(samples))]
    dicts = []
    for s in samples:
        d = ora.Dict.new(15, True)
        d.add_record(s)
        dicts.append(d)
    full = ora.Array.from_dicts(dicts, names)
    fk, fv, fc = full.export()
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    for p in parts:
        assert np.array_equal(p["rows"], fk["lo"].astype(np.int64))          # same global row set on every rank
        assert np.array_equal(p["present"], fc.astype(np.int32))              # reduced counts == full-matrix counts
        assert np.array_equal(p["unambig"], np.isin(fv, list(b"ACGT")).sum(axis=1))
    assert np.array_equal(np.concatenate([p["slab"] for p in parts], axis=1), fv)   # column slabs tile the matrix
    assert np.array_equal(parts[0]["mask"], parts[1]["mask"])
    want = np.zeros(len(fk), np.int32)
    for i, c in enumerate(b"-ACMTWYHGRSVKDBN"):
        if i:
            want |= (fv == c).any(axis=1).astype(np.int32) << i
    assert np.array_equal(parts[0]["mask"], want)                              # OR of the ranks' code sets == the full matrix's
    assert [(int(p["lo"]), int(p["hi"])) for p in parts] == [(0, 3), (3, 6)]


def test_shard_range_covers_everything():
    import dist as skdist
    for n in (1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            r = [skdist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))


# ---- all-vs-all distance over ranks -------------------------------------------------------------------------------------------
_PROB = {c: np.array(v, dtype=np.float64) for c, v in {
    ord("A"): [1, 0, 0, 0], ord("C"): [0, 1, 0, 0], ord("G"): [0, 0, 1, 0], ord("T"): [0, 0, 0, 1], ord("U"): [0, 0, 0, 1],
    ord("R"): [.5, 0, .5, 0], ord("Y"): [0, .5, 0, .5], ord("S"): [0, .5, .5, 0], ord("W"): [.5, 0, 0, .5], ord("K"): [0, 0, .5, .5],
    ord("M"): [.5, .5, 0, 0], ord("B"): [0, 1 / 3, 1 / 3, 1 / 3], ord("D"): [1 / 3, 0, 1 / 3, 1 / 3], ord("H"): [1 / 3, 1 / 3, 0, 1 / 3],
    ord("V"): [1 / 3, 1 / 3, 1 / 3, 0], ord("N"): [0, 0, 0, 0]}.items()}          # bit_encoding.rs:65-85 (N: all zero)


def _pair_numpy(cells, constant, filt_ambig, i_lo, i_hi):
    """merge_ska_array.rs:596-631 on a [S][U] byte matrix, pairs (i in [i_lo, i_hi), j > i)"""
    S = cells.shape[0]
    out = []
    for i in range(i_lo, i_hi):
        for j in range(i + 1, S):
            a, b = cells[i], cells[j]
            ga, gb = a == ord("-"), b == ord("-")
            mism = float((ga ^ gb).sum())
            both = ~ga & ~gb
            matches, distance = float(constant), 0.0
            if filt_ambig:
                un = both & np.isin(a, list(b"ACGTU")) & np.isin(b, list(b"ACGTU"))
                matches += float(un.sum())
                distance = float((un & (a != b)).sum())
            else:
                for x, y in zip(a[both], b[both]):
                    ov = float((_PROB[int(x)] * _PROB[int(y)]).sum())
                    if ov > 0:
                        matches += 1.0
                    distance += 1.0 - ov
            out.append((distance, 0.0 if matches + mism == 0 else mism / (matches + mism), float(int(matches)), float(int(mism))))
    return np.array(out, dtype=np.float64).reshape(-1, 4)


def _dist_worker(rank, world, port, tmp, filt_ambig, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dist as skdist
    d = np.load(os.path.join(tmp, "filtered.npz"))
    cells, constant = d["cells"], float(d["constant"])            # [S][U'] after generic_modes::distance's two filters
    lo, hi = skdist.shard_range(n, rank, world)
    slab = cells[lo:hi]
    U = slab.shape[1]
    # "planes" of the CPU stand-in: the cells themselves, eight per word (the engine exchanges 4 or 8 bit planes the same way)
    W = (U + 7) // 8
    packed = np.zeros((1, hi - lo, W * 8), dtype=np.uint8)
    packed[0, :, :U] = slab
    local = torch.from_numpy(packed.view(np.int64).reshape(1, hi - lo, W))

    def pair_fn(planes, i_lo, i_hi):
        c = planes.numpy().view(np.uint8).reshape(planes.shape[1], -1)[:, :U]
        return _pair_numpy(c, constant, filt_ambig, i_lo, i_hi)

    table = skdist.distance_sharded(local, pair_fn)
    if rank == 0:
        open(os.path.join(tmp, "sharded.tsv"), "wb").write(skdist.distance_tsv([f"s{i}" for i in range(n)], table))
    else:
        assert table is None
    dist.destroy_process_group()


@pytest.mark.parametrize("filt_ambig,world", [(True, 2), (False, 2), (True, 3)])
def test_sharded_distance_table_is_byte_identical(tmp_path, filt_ambig, world):
    """The pair matrix tiled over ranks (all-gather of per-rank planes, one band of rows per rank, gather of the finished pairs)
    gives the single-process `ska distance` table byte for byte (generic_modes.rs:136-189)."""
    import ora
    rng = np.random.default_rng(17)
    anc = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3000)
    n = 7
    dicts, names = [], [f"s{i}" for i in range(n)]
    for i in range(n):
        s = anc.copy()
        pos = rng.integers(0, len(s), size=40)
        s[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=40)
        rec = bytes(s[: len(s) - 150 * (i % 3)].tolist())
        d = ora.Dict.new(9, True)                                  # k = 9 on 3 kb: repeats -> ambiguity codes
        d.add_record(rec)
        d.add_record(rec[200:500][::-1])
        dicts.append(d)
    full = ora.Array.from_dicts(dicts, names)
    want = ora.Array.from_dicts(dicts, names).distance_tsv(0.5, filt_ambig)
    # the filtered matrix every rank's slab comes from, and the constant-site count (generic_modes.rs:149-168)
    full.apply_filters(0.5, False, ora.FILTER_NONE, False, False)
    constant = full.apply_filters(0.0, False, ora.FILTER_NO_CONST, False, False)
    cells = np.ascontiguousarray(full.export()[1].T)
    np.savez(os.path.join(str(tmp_path), "filtered.npz"), cells=cells, constant=constant)
    port = 29700 + (os.getpid() % 500) + 7 * world + (3 if filt_ambig else 0)
    mp.spawn(_dist_worker, args=(world, port, str(tmp_path), filt_ambig, n), nprocs=world, join=True)
    assert open(os.path.join(str(tmp_path), "sharded.tsv"), "rb").read() == want


def test_pair_bands_partition_the_pair_matrix():
    import dist as skdist
    for n in (2, 7, 33, 100, 1000, 8000):
        for w in (1, 2, 3, 8):
            b = skdist.pair_bands(n, w)
            assert len(b) == w and b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(lo % 8 == 0 for lo, hi in b if lo < hi)
            if n >= 1000 and w == 8:                                   # balanced by pairs, not by rows
                pairs = [sum(n - 1 - i for i in range(lo, hi)) for lo, hi in b]
                assert max(pairs) < 1.1 * (n * (n - 1) // 2) / w, (n, pairs)


def test_row_stat_reduction_edges():
    """codes with bit 15 ('N') and counts up to the 15-bit packing limit survive the exchange (world of one: the arithmetic)"""
    import dist as skdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        for total in (0x7FFF, 0xFFFF, None):
            present = torch.tensor([0, 1, 0x7FFF, 12345], dtype=torch.int32)
            unambig = torch.tensor([0, 1, 0x7FFF, 12000], dtype=torch.int32)
            mask = torch.tensor([0, 1 << 15, 0xFFFE, (1 << 15) | 2], dtype=torch.int32)
            p, u, m = skdist.reduce_row_stats(present.clone(), unambig.clone(), mask.clone(), total_samples=total)
            assert torch.equal(p, present) and torch.equal(u, unambig) and torch.equal(m, mask)
    finally:
        dist.destroy_process_group()
