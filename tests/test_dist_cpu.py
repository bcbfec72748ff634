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
