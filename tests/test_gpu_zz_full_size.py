"""BASELINE.json configs[1] and configs[2] at their full sizes (`-m gpu`): 100 and 1 000 synthetic 5 Mbp assemblies, k = 31.
The oracle cannot build 1 000 samples in test time, so config 3 is pinned by (a) oracle dictionaries of samples spread over the
set, (b) the oracle's merged array on a 64-sample subset built from the same files, (c) exact sharded == unsharded equality of the
whole array on the device, (d) size-independent properties of the filter / alignment.  Config 2 goes through the `ska` executable:
its 100-sample .skf is read by the oracle and must be the oracle's own array, row for row."""
import os
import subprocess

import numpy as np
import pytest

import ora

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")


@pytest.fixture(scope="module")
def E():
    import torch                    # before the engine: torch brings its own copy of the HIP runtime, and the copy loaded second finds no GPU
    torch.cuda.init()
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def _shm(tmp_path_factory):
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    import tempfile
    return tempfile.mkdtemp(prefix="skx_full_", dir=base)


def _sorted(arr):
    k, v, c = arr.export()
    o = np.argsort(k["lo"], kind="stable")
    return k["lo"][o], v[o], c[o]


@pytest.fixture(scope="module")
def thousand(tmp_path_factory):
    import shutil
    import synth
    td = _shm(tmp_path_factory)
    anc = synth.ancestor(5_000_000, seed=1)
    n = 1000
    files = [os.path.join(td, f"g{i}.fa") for i in range(n)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as pool:                 # numpy's copies and the file writes release the interpreter lock
        list(pool.map(lambda i: synth.to_fasta(synth.sample_stream(anc, i, n), files[i]), range(n)))
    yield td, files
    shutil.rmtree(td, ignore_errors=True)


def test_config3_thousand_assemblies(E, thousand):
    import torch
    import dist as skdist
    td, files = thousand
    n = len(files)
    names = [f"g{i}" for i in range(n)]
    dev = torch.device("cuda", 0)
    whole_ds = E.DictSet.from_files([(f, None) for f in files], 31, True, threads=32)
    # (a) dictionaries of 8 samples spread over the set (index 9 mod 10 = reverse-complemented by the generator)
    for i in (0, 9, 199, 333, 500, 509, 777, 999):
        ok, ob = ora.Dict.from_files(31, files[i]).export()
        gk, gb = whole_ds.export(i)
        assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gb, ob), i
    whole = whole_ds.merge(names)
    U = whole.nrows
    assert 20_000_000 < U < 26_000_000
    assert [int(x) for x in whole.sample_kmers()] == [whole_ds.size(i) for i in range(n)]
    # (b) the oracle's merged array on a 64-sample subset (every 16th sample, so the subset spans the whole tree), same files
    sub = list(range(0, n, 16))[:64]
    sub_inputs = [(names[i], files[i], None) for i in sub]
    oa = ora.Array.build(sub_inputs, k=31, threads=8)
    ga = E.Array.build(sub_inputs, k=31, threads=16)
    for x, y in zip(_sorted(ga), _sorted(oa)):
        assert np.array_equal(x, y)
    # ... and the whole array restricted to those samples is that array: rows any of them has, in key order
    wkeys, wcounts = whole.export_keys()
    p, pitch, rows = whole.device_matrix()
    assert rows == U
    mat = skdist.as_tensor(p, n * pitch, "|u1", dev).view(n, pitch)[:, :U]
    sub_mat = mat[torch.tensor(sub, device=dev)]                                    # [64][U], engine (hash) order
    present = (sub_mat != ord("-")).any(dim=0)
    assert len(wkeys) == U and int(wcounts.sum()) == sum(whole_ds.size(i) for i in range(n))
    # the two arrays hold the same rows: compared as multisets of rows through a 64-bit row hash (the whole array is in hash
    # order, the oracle's in key order)
    sub_rows = sub_mat[:, present].t().contiguous().cpu().numpy()                   # [rows of the subset][samples of the subset]
    w = np.random.default_rng(5).integers(1, 1 << 62, size=len(sub), dtype=np.uint64) | np.uint64(1)
    ha = np.sort(sub_rows.astype(np.uint64) @ w)
    hb = np.sort(_sorted(oa)[1].astype(np.uint64) @ w)
    assert len(ha) == len(hb) and np.array_equal(ha, hb)
    # (c) sharded == unsharded, exactly: two shards exchange key tables, fill their column slabs; slabs tile the whole matrix
    half = n // 2
    shards = [E.DictSet.from_files([(f, None) for f in files[lo:hi]], 31, True, threads=32) for lo, hi in ((0, half), (half, n))]
    rows_ks = E.KeySet.merge([s.union_keys() for s in shards])
    assert len(rows_ks) == U
    wp = [skdist.as_tensor(x, U, "<i4", dev) for x in whole.device_stats()[:3]]
    acc = [torch.zeros(U, dtype=torch.int32, device=dev) for _ in range(2)]
    accm = torch.zeros(U, dtype=torch.int32, device=dev)
    for s, (lo, hi) in zip(shards, ((0, half), (half, n))):
        part = s.assemble(rows_ks, names[lo:hi])
        pp, ppitch, prow = part.device_matrix()
        assert prow == U
        pm = skdist.as_tensor(pp, (hi - lo) * ppitch, "|u1", dev).view(hi - lo, ppitch)[:, :U]
        assert torch.equal(pm, mat[lo:hi])
        st = [skdist.as_tensor(x, U, "<i4", dev) for x in part.device_stats()[:3]]
        acc[0] += st[0]; acc[1] += st[1]; accm |= st[2]
        part.free(); s.free()
    assert torch.equal(acc[0], wp[0]) and torch.equal(acc[1], wp[1]) and torch.equal(accm, wp[2])
    # (d) the default `ska align` filter at full size: idempotent, every kept column varies and is present in >= 900 samples
    removed = whole.apply_filters(0.9)
    kept = whole.nrows
    assert removed == U - kept and 4_000_000 < kept < 6_000_000
    assert whole.apply_filters(0.9) == 0 and whole.nrows == kept
    p, pitch, rows = whole.device_matrix()
    fm = skdist.as_tensor(p, n * pitch, "|u1", dev).view(n, pitch)[:, :kept]
    assert bool((fm != fm[0]).any(dim=0).all())
    assert bool(((fm != ord("-")).sum(dim=0) >= 900).all())
    whole_ds.free()


def test_config2_hundred_assemblies_through_the_executable(E, thousand):
    td, files = thousand
    import synth
    # config 2 is its own data set (100 samples: a shallower tree than the first 100 of 1 000)
    anc = synth.ancestor(5_000_000, seed=1)
    n = 100
    sub = os.path.join(td, "c2")
    os.makedirs(sub, exist_ok=True)
    fl = []
    for i in range(n):
        p = os.path.join(sub, f"h{i}.fa")
        synth.to_fasta(synth.sample_stream(anc, i, n), p)
        fl.append(p)
    with open(os.path.join(sub, "list.txt"), "w") as f:
        for i, p in enumerate(fl):
            f.write(f"h{i}\t{p}\n")
    r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", "c2", "-k", "31", "--threads", "32"], cwd=sub, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    got = ora.Array.load(os.path.join(sub, "c2.skf"))                       # the oracle reads the engine's file
    want = ora.Array.build([(f"h{i}", p, None) for i, p in enumerate(fl)], k=31, threads=8)
    assert got.names == want.names and got.k == 31 and got.rc and got.version == "0.5.2"
    for x, y in zip(_sorted(got), _sorted(want)):
        assert np.array_equal(x, y)
    # and `ska align` of that file is the oracle's alignment, column for column
    r = subprocess.run([SKA, "align", "c2.skf", "-o", "c2.aln"], cwd=sub, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    g = open(os.path.join(sub, "c2.aln"), "rb").read()
    o = want.align(min_freq=0.9)

    def cols(aln):
        rws = aln.split(b"\n")[1::2]
        m = np.frombuffer(b"".join(rws), dtype=np.uint8).reshape(len(rws), -1)
        return m[:, np.lexsort(m[::-1])]
    assert g.split(b"\n")[0::2][:n] == o.split(b"\n")[0::2][:n]
    assert np.array_equal(cols(g), cols(o))


def test_config5_one_isolate_at_size(E, isolates):
    """BASELINE.json configs[4] shape, one isolate at its real size: 2 x 150 bp reads at 50x of a 5 Mbp genome (833 334 pairs, 252 M
    stream bytes), 0.5 % substitution errors, per-cycle Phred profile (synth.write_read_pair: isolate 0 of the module's read sets);
    k = 41 (128-bit keys), --min-count 5, --qual-filter strict --min-qual 20.  The engine's own partition kernels (skx_reads2.hip)
    against the oracle's sequential KmerFilter, exactly.  (The oracle's three dictionaries are computed on threads of their own: 20 s
    each on one core.)"""
    from concurrent.futures import ThreadPoolExecutor
    files = list(isolates[2][0])
    qo = ora.qual(5, 20, ora.QUAL_STRICT)
    pool = ThreadPoolExecutor(max_workers=3)
    want = {k: pool.submit(lambda k=k: ora.Dict.from_files(k, files[0], files[1], True, qo).export()) for k in (41, 31, 17)}
    ok, ob = want[41].result()
    ds = E.DictSet.from_files([(files[0], files[1])], 41, True, E.qual(5, 20, E.QUAL_STRICT), threads=1)
    gk, gb = ds.export(0)
    assert 4_900_000 < len(gk) < 5_100_000
    assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)
    # the sort-based first form (rocPRIM) agrees as well
    os.environ["SKX_KNOBS"] = "reads_sort=1"
    try:
        ds2 = E.DictSet.from_files([(files[0], files[1])], 41, True, E.qual(5, 20, E.QUAL_STRICT), threads=1)
    finally:
        os.environ.pop("SKX_KNOBS", None)
    k2, b2 = ds2.export(0)
    assert np.array_equal(k2["lo"], gk["lo"]) and np.array_equal(k2["hi"], gk["hi"]) and np.array_equal(b2, gb)
    # smaller k = more gated windows per read: k = 31 runs the 48-word partitions at their tighter head-room, k = 17 (the reference's
    # default) the 24-word partitions -- both against the oracle's sequential filter as well
    for k in (31, 17):
        ok, ob = want[k].result()
        dsk = E.DictSet.from_files([(files[0], files[1])], k, True, E.qual(5, 20, E.QUAL_STRICT), threads=1)
        gk, gb = dsk.export(0)
        assert len(gk) > 4_000_000 and np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob), k


@pytest.fixture(scope="module")
def isolates(tmp_path_factory):
    """eight read sets of BASELINE config 5's shape at their real size (2 x 150 bp at 50x of a 5 Mbp genome each: 2 x 126 MB of FASTQ)"""
    import shutil
    import synth
    from concurrent.futures import ProcessPoolExecutor
    td = _shm(tmp_path_factory)
    n = 8
    import multiprocessing
    with ProcessPoolExecutor(max_workers=min(n, os.cpu_count() or 1), mp_context=multiprocessing.get_context("spawn")) as ex:   # not a fork of a process that holds the GPU
        pairs = list(ex.map(synth.write_read_pair_of, range(n), [n] * n, [os.path.join(td, f"iso{i}") for i in range(n)]))
    lst = os.path.join(td, "list.txt")
    with open(lst, "w") as f:
        for i, (a, b) in enumerate(pairs):
            f.write(f"iso{i}\t{a}\t{b}\n")
    yield td, [f"iso{i}" for i in range(n)], pairs, lst
    shutil.rmtree(td, ignore_errors=True)


def test_config5_flow_eight_isolates_at_size(E, isolates):
    """BASELINE.json configs[4] as a flow on one GPU: paired FASTQ isolates -> `ska build -k 41 --min-count 5 --min-qual 20 --qual-filter
    strict` -> .skf -> `ska distance` (both modes, with and without --min-freq) and `ska align`, against the oracle's build_and_merge
    (ska_dict.rs:118-180 + bloom_filter.rs:116-148 per isolate, merge_ska_dict.rs append/merge) and generic_modes::distance / align
    on the same files: the .skf read by the oracle is the oracle's own array row for row, the tables are byte-identical, the alignment
    has the oracle's columns.  The sharded form (`--gpus 2`, ranks sharing the GPU) writes the same bytes."""
    td, names, pairs, lst = isolates
    opts = ["-k", "41", "--min-count", "5", "--min-qual", "20", "--qual-filter", "strict"]
    r = subprocess.run([SKA, "build", "-f", lst, "-o", "c5", "--threads", "16", *opts], cwd=td, capture_output=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-800:]
    # (the oracle's dictionaries on one thread each, then its append / merge and MergeSkaArray::new: build_and_merge itself takes one thread
    # for eight samples -- merge_ska_dict.rs:384 -- and 20 s per isolate)
    from concurrent.futures import ThreadPoolExecutor
    qo = ora.qual(5, 20, ora.QUAL_STRICT)
    with ThreadPoolExecutor(max_workers=len(pairs)) as pool:
        dicts = list(pool.map(lambda ab: ora.Dict.from_files(41, ab[0], ab[1], True, qo), pairs))
    want = ora.Array.from_dicts(dicts, names)
    got = ora.Array.load(os.path.join(td, "c5.skf"))                  # the engine's file through the oracle's reader
    assert got.names == names
    got.sort_rows(); want.sort_rows()
    gk, gv, gc = got.export()
    ok, ov, oc = want.export()
    assert 5_000_000 < len(ok) < 6_500_000                             # 128-bit keys: the u128 path
    assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gv, ov) and np.array_equal(gc, oc)
    want.save(os.path.join(td, "oracle.skf"))                          # the oracle's filters work in place: a fresh copy per table
    for flags, mf, filt in (([], 0.0, True), (["--allow-ambiguous"], 0.0, False), (["--min-freq", "0.9"], 0.9, True)):
        r = subprocess.run([SKA, "distance", "c5.skf", "-o", "c5.tsv", *flags], cwd=td, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        tsv = open(os.path.join(td, "c5.tsv"), "rb").read()
        assert tsv == ora.Array.load(os.path.join(td, "c5.skf")).distance_tsv(mf, filt), flags
        if filt:                                                        # integer counts: independent of the row order
            assert tsv == ora.Array.load(os.path.join(td, "oracle.skf")).distance_tsv(mf, filt), flags
    # (the oracle array the tables come from is the file's: it was shown equal to the oracle's own build above, and rows in file order
    # keep the reference's f64 accumulation order out of the comparison for --allow-ambiguous)
    r = subprocess.run([SKA, "align", "c5.skf", "-o", "c5.aln"], cwd=td, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    g = open(os.path.join(td, "c5.aln"), "rb").read()
    assert g == ora.Array.load(os.path.join(td, "c5.skf")).align(min_freq=0.9)          # file order kept: byte-identical
    o = ora.Array.load(os.path.join(td, "oracle.skf")).align(min_freq=0.9)

    def cols(aln):
        rws = aln.split(b"\n")[1::2]
        m = np.frombuffer(b"".join(rws), dtype=np.uint8).reshape(len(rws), -1)
        return m[:, np.lexsort(m[::-1])]
    assert g.split(b"\n")[0::2] == o.split(b"\n")[0::2] and np.array_equal(cols(g), cols(o))
    # two ranks sharing the GPU, exchanges through skx_comm_* (host-staged transport)
    env = dict(os.environ, SKX_COMM="local", SKX_DEVICE="0")
    r = subprocess.run([SKA, "distance", "--gpus", "2", "-f", lst, "-o", "c5_2.tsv", "--threads", "8", *opts], cwd=td, capture_output=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-800:]
    r = subprocess.run([SKA, "distance", "c5.skf", "-o", "c5.tsv"], cwd=td, capture_output=True, timeout=600)
    assert open(os.path.join(td, "c5_2.tsv"), "rb").read() == open(os.path.join(td, "c5.tsv"), "rb").read()


def test_config4_one_ranks_share(E):
    """BASELINE.json configs[3] (8 000 assemblies over 8 GPUs, 100 private SNPs each), one rank's share on one GPU: the eight ranks'
    key tables are built one after the other (each from its own 1 000 samples), united as skx_keyset_allgather unites them, and rank
    0 fills its column slab over the ~50 M global rows.  The row set is checked against an independent union (torch.unique), columns
    against the oracle's dictionaries of their samples, the slab's statistics against the columns."""
    import torch
    import synth
    import dist as skdist
    world, G = 8, 1000
    n_total = world * G
    dev = torch.device("cuda", 0)
    anc = synth.ancestor(5_000_000, seed=1)
    tables, ds0, streams0 = [], None, {}
    for r in range(world):
        streams = [synth.sample_stream(anc, r * G + i, n_total, private_snps=100).tobytes() for i in range(G)]
        if r == 0:
            streams0 = {i: streams[i] for i in (0, 9, 500, 999)}
        ds = E.DictSet.build(streams, 31, True)
        del streams
        ks = ds.union_keys()
        p, n, _ = ks.device()
        tables.append(skdist.as_tensor(p, n, "<i8", dev).clone())
        ks.free()
        if r == 0:
            ds0 = ds
        else:
            ds.free()
    sets = [E.KeySet.from_device(t.data_ptr(), t.numel(), 31, True) for t in tables]
    rows = E.KeySet.merge(sets)
    U = len(rows)
    assert 40_000_000 < U < 60_000_000
    p, n, _ = rows.device()
    got_rows = skdist.as_tensor(p, n, "<i8", dev)
    want_rows = torch.unique(torch.cat(tables))                         # signed order != engine order: compare as sorted sets
    assert want_rows.numel() == U and torch.equal(torch.sort(got_rows).values, want_rows)
    del want_rows, tables, sets
    arr = ds0.assemble(rows, [f"g{i}" for i in range(G)])              # 50 GB: the eager slab of rank 0
    arr.set_total_samples(n_total)
    assert arr.nrows == U and arr.nsamples == G
    keys, counts = arr.export_keys()                                    # sorted by key, with the slab's per-row counts
    assert int(counts.sum()) == sum(ds0.size(i) for i in range(G))
    # columns against the oracle's dictionaries (index 9 is reverse-complemented by the generator)
    pm, pitch, nrows = arr.device_matrix()
    mat = skdist.as_tensor(pm, G * pitch, "|u1", dev).view(G, pitch)[:, :U]
    pp = skdist.as_tensor(arr.device_stats()[0], U, "<i4", dev)
    for c0 in range(0, U, 1 << 21):                                     # 50 GB of cells: a slice of rows at a time
        c1 = min(U, c0 + (1 << 21))
        assert torch.equal((mat[:, c0:c1] != ord("-")).sum(dim=0, dtype=torch.int32), pp[c0:c1]), c0
    for i, s in streams0.items():
        d = ora.Dict.new(31, True)
        for rec in s.split(b"\n")[:-1]:
            d.add_record(rec)
        ok, ob = d.export()
        gk, gb = ds0.export(i)
        assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gb, ob), i
        assert int((mat[i] != ord("-")).sum()) == len(ok), i            # the column holds exactly the sample's split k-mers
        assert np.isin(ok["lo"], keys["lo"], assume_unique=True).all(), i
    arr.free(); ds0.free()


def test_config5_batched_build_merge_distance_chain():
    """tools/reads_1000.py in small: BASELINE.json configs[4]'s chain as a user with more isolates than one build holds runs it -- paired
    FASTQ isolates (2 x 150 bp at 50x of a 500 kbp genome; the last batch as .fastq.gz) in 4 batches of 8 through `ska build -k 41
    --min-count 5 --min-qual 20 --qual-filter strict`, the batch files joined by `ska merge` (merge_ska_dict.rs:160-193 through
    generic_modes.rs / lib.rs:728-741), then `ska distance` over all 496 pairs.  Two isolates of every batch are checked against the
    oracle: the isolate's column of its batch .skf is the oracle's dictionary of its two files (ska_dict.rs:118-180,
    bloom_filter.rs:116-148), and the rows of the final table for every pair of those eight isolates are the rows of the oracle's table
    over an array of its own dictionaries (generic_modes.rs:136-189: a pair's counts do not depend on the other samples)."""
    import multiprocessing
    import shutil
    import tempfile
    import zlib
    from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
    import synth
    td = tempfile.mkdtemp(prefix="skx_c5chain_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        N, NB, G = 32, 4, 500_000
        opts = ["-k", "41", "--min-count", "5", "--min-qual", "20", "--qual-filter", "strict"]
        qo = ora.qual(5, 20, ora.QUAL_STRICT)
        spot_dicts, spot_batch = {}, {}
        with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1), mp_context=multiprocessing.get_context("spawn")) as ex, ThreadPoolExecutor(max_workers=2) as opool:
            for b in range(NB):
                lo, hi = N * b // NB, N * (b + 1) // NB
                pairs = list(ex.map(synth.write_read_pair_of, range(lo, hi), [N] * (hi - lo), [os.path.join(td, f"iso{i}") for i in range(lo, hi)], [G] * (hi - lo)))
                if b == NB - 1:                                     # what read sets come as
                    gz = []
                    for pr in pairs:
                        out = []
                        for f in pr:
                            c = zlib.compressobj(1, zlib.DEFLATED, 31)
                            with open(f, "rb") as src, open(f + ".gz", "wb") as dst:
                                dst.write(c.compress(src.read()) + c.flush())
                            os.unlink(f)
                            out.append(f + ".gz")
                        gz.append(tuple(out))
                    pairs = gz
                with open(os.path.join(td, f"list{b}.txt"), "w") as lst:
                    for i, (f1, f2) in zip(range(lo, hi), pairs):
                        lst.write(f"iso{i}\t{f1}\t{f2}\n")
                r = subprocess.run([SKA, "build", "-f", f"list{b}.txt", "-o", f"batch{b}", "--threads", "8", *opts], cwd=td, capture_output=True, timeout=900)
                assert r.returncode == 0, r.stderr[-800:]
                spots = (lo, hi - 1)
                ds = list(opool.map(lambda i: ora.Dict.from_files(41, pairs[i - lo][0], pairs[i - lo][1], True, qo), spots))
                for i, d in zip(spots, ds):
                    spot_dicts[i] = d
                    spot_batch[i] = (b, lo)
                for pr in pairs:
                    for f in pr:
                        os.unlink(f)
        for i, (b, lo) in sorted(spot_batch.items()):
            keys, var, _ = ora.Array.load(os.path.join(td, f"batch{b}.skf")).export()
            order = np.lexsort((keys["lo"], keys["hi"]))
            ok, ob = spot_dicts[i].export()
            col = var[order, i - lo]
            have = col != ord("-")
            assert 400_000 < len(ok) < 700_000
            assert int(have.sum()) == len(ok) and np.array_equal(keys["lo"][order][have], ok["lo"]) and np.array_equal(keys["hi"][order][have], ok["hi"]), i
            assert np.array_equal(col[have], ob), i
        r = subprocess.run([SKA, "merge", *[f"batch{b}.skf" for b in range(NB)], "-o", "all"], cwd=td, capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr[-800:]
        merged = ora.Array.load(os.path.join(td, "all.skf"))
        assert merged.names == [f"iso{i}" for i in range(N)]
        r = subprocess.run([SKA, "distance", "all.skf", "-o", "all.tsv"], cwd=td, capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr[-800:]
        tsv = open(os.path.join(td, "all.tsv"), "rb").read()
        assert tsv == merged.distance_tsv(0.0, True)                # the oracle's table over the merged file: all 496 rows
        order = sorted(spot_dicts)
        oarr = ora.Array.from_dicts([spot_dicts[i] for i in order], [f"iso{i}" for i in order])
        want = {tuple(l.split("\t")[:2]): l for l in oarr.distance_tsv(0.0, True).decode().splitlines()[1:]}
        got = {tuple(l.split("\t")[:2]): l for l in tsv.decode().splitlines()[1:]}
        assert len(got) == N * (N - 1) // 2 and len(want) == 28
        assert all(got.get(k) == v for k, v in want.items()), [k for k, v in want.items() if got.get(k) != v][:3]
    finally:
        shutil.rmtree(td, ignore_errors=True)
