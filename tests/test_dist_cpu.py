"""World-size-2/3 tests on CPU of the multi-GPU exchanges behind the C ABI (include/skx.h "Collectives", csrc/skx_comm.hip) through
their host-staged transport on host buffers -- no GPU involved -- with torch.distributed / gloo as the launcher's rendezvous and as
an independent checker of the primitives (skx_comm_allgather == gloo all_gather, skx_comm_allreduce_u32 == gloo all_reduce).

The data are key tables / statistics / cells produced by the CPU oracle (the checker), and the ranks do with numpy exactly what
the engine's exchanges do on the device (skx_keyset_allgather: sizes -> padded tables -> one all-gather -> union;
skx_array_reduce_stats: packed counts in one all-reduce, 16-bit code sets all-gathered and OR-ed; skx_array_distance_sharded:
planes all-gathered, bands of skx_pair_bands, finished pairs gathered on rank 0), so the exchange is verified end to end: union of
the gathered per-shard tables == the row set of one merge over all samples, reduced statistics == statistics of the full matrix,
sharded distance table byte-identical to the oracle's."""
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


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dist as skdist
    return skdist, skdist.make_comm(None, rank, world, transport="local")     # ctx None: host buffers only


def _allgather_padded(comm, table):
    """what skx_keyset_allgather does: sizes, tables padded to the longest, one all-gather"""
    sizes = comm.allgather_host(np.array([len(table)], np.uint64)).reshape(-1)
    mx = max(int(sizes.max()), 1)
    padded = np.zeros(mx, table.dtype)
    padded[: len(table)] = table
    got = comm.allgather_host(padded)
    return [got[r, : int(sizes[r])] for r in range(comm.world)]


def _worker(rank, world, port, tmp, packed):
    skdist, comm = _init(rank, world, port)
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
    tables = _allgather_padded(comm, lk["lo"].astype(np.uint64))
    # gloo as the independent checker of the primitive
    mine = torch.from_numpy(np.resize(lk["lo"].astype(np.int64), 64))
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    assert np.array_equal(comm.allgather_host(mine.numpy()), torch.stack(parts).numpy())
    rows = np.unique(np.concatenate(tables))
    # this rank's column slab on the global rows
    idx = np.searchsorted(rows, lk["lo"].astype(np.uint64))
    slab = np.full((len(rows), hi - lo), ord("-"), dtype=np.uint8)
    slab[idx] = lv
    present = (slab != ord("-")).sum(axis=1).astype(np.uint32)
    unambig = np.isin(slab, list(b"ACGT")).sum(axis=1).astype(np.uint32)
    codes = {c: i for i, c in enumerate(b"-ACMTWYHGRSVKDBN")}
    mask = np.zeros(len(rows), dtype=np.uint32)
    for c, i in codes.items():
        if i:
            mask |= ((slab == c).any(axis=1).astype(np.uint32) << i)
    # skx_array_reduce_stats: both counts in one all-reduce while the job has < 32 768 samples, else two; code sets as 16-bit words
    if packed:
        both = comm.allreduce_u32_host(present + (unambig << 16))
        g_present, g_unambig = both & 0xFFFF, both >> 16
    else:
        g_present, g_unambig = comm.allreduce_u32_host(present), comm.allreduce_u32_host(unambig)
    t = torch.from_numpy(present.astype(np.int64))
    dist.all_reduce(t)
    assert np.array_equal(g_present, t.numpy())
    g_mask = np.bitwise_or.reduce(comm.allgather_host(mask.astype(np.uint16)), axis=0).astype(np.uint32)
    np.savez(os.path.join(tmp, f"r{rank}.npz"), rows=rows, present=g_present, unambig=g_unambig, mask=g_mask, lo=lo, hi=hi, slab=slab)
    comm.barrier()
    comm.free()
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
        assert np.array_equal(p["rows"], fk["lo"].astype(np.uint64))          # same global row set on every rank
        assert np.array_equal(p["present"], fc.astype(np.uint32))             # reduced counts == full-matrix counts
        assert np.array_equal(p["unambig"], np.isin(fv, list(b"ACGT")).sum(axis=1))
    assert np.array_equal(np.concatenate([p["slab"] for p in parts], axis=1), fv)   # column slabs tile the matrix
    assert np.array_equal(parts[0]["mask"], parts[1]["mask"])
    want = np.zeros(len(fk), np.uint32)
    for i, c in enumerate(b"-ACMTWYHGRSVKDBN"):
        if i:
            want |= (fv == c).any(axis=1).astype(np.uint32) << i
    assert np.array_equal(parts[0]["mask"], want)                              # OR of the ranks' code sets == the full matrix's
    assert [(int(p["lo"]), int(p["hi"])) for p in parts] == [(0, 3), (3, 6)]


def test_shard_range_covers_everything():
    import dist as skdist
    for n in (1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            r = [skdist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


# ---- all-vs-all distance over ranks -------------------------------------------------------------------------------------------
_PROB = {c: np.array(v, dtype=np.float64) for c, v in {
    ord("A"): [1, 0, 0, 0], ord("C"): [0, 1, 0, 0], ord("G"): [0, 0, 1, 0], ord("T"): [0, 0, 0, 1], ord("U"): [0, 0, 0, 1],
    ord("R"): [.5, 0, .5, 0], ord("Y"): [0, .5, 0, .5], ord("S"): [0, .5, .5, 0], ord("W"): [.5, 0, 0, .5], ord("K"): [0, 0, .5, .5],
    ord("M"): [.5, .5, 0, 0], ord("B"): [0, 1 / 3, 1 / 3, 1 / 3], ord("D"): [1 / 3, 0, 1 / 3, 1 / 3], ord("H"): [1 / 3, 1 / 3, 0, 1 / 3],
    ord("V"): [1 / 3, 1 / 3, 1 / 3, 0], ord("N"): [0, 0, 0, 0]}.items()}          # bit_encoding.rs:65-85 (N: all zero)


def _pair_numpy(cells, constant, filt_ambig, i_lo, i_hi):
    """merge_ska_array.rs:596-631 on a [S][U] byte matrix, pairs (i in [i_lo, i_hi), j > i)"""
    import skx_engine as E
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
            out.append((distance, 0.0 if matches + mism == 0 else mism / (matches + mism), int(matches), int(mism)))
    return np.array(out, dtype=E.DIST_DT)


def _dist_worker(rank, world, port, tmp, filt_ambig, n):
    skdist, comm = _init(rank, world, port)
    import skx_engine as E
    d = np.load(os.path.join(tmp, "filtered.npz"))
    cells, constant = d["cells"], float(d["constant"])            # [S][U'] after generic_modes::distance's two filters
    lo, hi = skdist.shard_range(n, rank, world)
    slab = cells[lo:hi]
    U = slab.shape[1]
    # skx_array_distance_sharded's steps with the cells themselves as the "planes" of the CPU stand-in: sizes, padded to the largest
    # shard, one all-gather, samples back in rank order
    sizes = comm.allgather_host(np.array([hi - lo], np.uint64)).reshape(-1)
    mx = int(sizes.max())
    padded = np.zeros((mx, U), np.uint8)
    padded[: hi - lo] = slab
    got = comm.allgather_host(padded)
    planes = np.concatenate([got[r, : int(sizes[r])] for r in range(world)], axis=0)
    assert np.array_equal(planes, cells)
    bands = skdist.pair_bands(n, world)
    b_lo, b_hi = bands[rank]
    mine = _pair_numpy(planes, constant, filt_ambig, b_lo, b_hi)
    counts = [sum(n - 1 - i for i in range(a, b)) for a, b in bands]
    assert len(mine) == counts[rank]
    table = comm.gather_root_host(mine, [c * E.DIST_DT.itemsize for c in counts])
    if rank == 0:
        open(os.path.join(tmp, "sharded.tsv"), "wb").write(skdist.distance_tsv([f"s{i}" for i in range(n)], table.view(E.DIST_DT)))
    else:
        assert table is None
    comm.free()
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


def _edge_worker(rank, world, port, tmp):
    skdist, comm = _init(rank, world, port)
    # counts up to the 15-bit packing limit and code sets with bit 15 ('N') survive the exchange; zero-length and ragged blocks
    present = np.array([0, 1, 0x7FFF if rank == 0 else 0, 12345], np.uint32)
    unambig = np.array([0, 1, 0x7FFF if rank == 0 else 0, 12000], np.uint32)
    both = comm.allreduce_u32_host(present + (unambig << 16))
    assert list(both & 0xFFFF) == [0, world, 0x7FFF, 12345 * world] and list(both >> 16) == [0, world, 0x7FFF, 12000 * world]
    m = np.array([0, 1 << 15, 0xFFFE if rank == 1 else 0, (1 << 15) | (2 << rank)], np.uint16)
    g = np.bitwise_or.reduce(comm.allgather_host(m), axis=0)
    assert list(g) == [0, 1 << 15, 0xFFFE, (1 << 15) | sum(2 << r for r in range(world))]
    assert comm.allgather_host(np.zeros(0, np.uint8)).shape == (world, 0)
    blocks = [bytes([r]) * (5 * r) for r in range(world)]                # rank 0 contributes nothing
    got = comm.gather_root_host(np.frombuffer(blocks[rank], np.uint8), [len(b) for b in blocks])
    if rank == 0:
        assert got.tobytes() == b"".join(blocks)
    for _ in range(50):
        comm.barrier()
    comm.free()
    dist.destroy_process_group()


def test_row_stat_reduction_edges_and_ragged_gather(tmp_path):
    mp.spawn(_edge_worker, args=(3, 29950 + os.getpid() % 40, str(tmp_path)), nprocs=3, join=True)


def test_comm_world_of_one_needs_no_peer(tmp_path):
    import skx_engine as E
    c = E.Comm.local(0, 1, str(tmp_path))
    a = np.arange(7, dtype=np.uint32)
    assert np.array_equal(c.allgather_host(a)[0], a) and np.array_equal(c.allreduce_u32_host(a), a)
    c.barrier()
    c.free()
