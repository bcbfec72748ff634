"""Differential parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs (`-m gpu`).
Bit-exact: integer split k-mers, byte middle bases, integer counts; distances compared as printed."""
import os

import numpy as np
import pytest
from conftest import set_knob, del_knob

import golden_cases as G
import ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def rand_records(rng, n_rec, max_len, alphabet=b"ACGT", p_bad=0.0, lower=False):
    recs = []
    for _ in range(n_rec):
        L = int(rng.integers(0, max_len + 1))
        s = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=L)
        if p_bad > 0 and L:
            bad = rng.random(L) < p_bad
            s = np.where(bad, rng.choice(np.frombuffer(b"NnRYKM-.*", dtype=np.uint8), size=L), s)
        if lower and L:
            s = np.where(rng.random(L) < 0.2, s | 0x20, s)
        recs.append(s.astype(np.uint8).tobytes())
    return recs


def oracle_dict(recs, k, rc):
    d = ora.Dict.new(k, rc)
    for r in recs:
        d.add_record(r)
    return d


def check_dicts(E, samples, k, rc):
    ds = E.DictSet.build([E.record_stream(r) for r in samples], k, rc)
    for i, recs in enumerate(samples):
        ok, ob = oracle_dict(recs, k, rc).export()
        gk, gb = ds.export(i)
        assert len(gk) == len(ok), (k, rc, i, len(gk), len(ok))
        assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"])
        assert np.array_equal(gb, ob)
    return ds


@pytest.mark.parametrize("k", [5, 7, 9, 15, 17, 21, 31, 33, 41, 55, 63])
@pytest.mark.parametrize("rc", [True, False])
def test_dict_random(E, k, rc):
    rng = np.random.default_rng(100 + k)
    samples = [rand_records(rng, 6, 3000) + [b"ACGT" * 3] for _ in range(3)]
    check_dicts(E, samples, k, rc)


@pytest.mark.parametrize("k", [7, 15, 31, 33, 47, 63])
def test_dict_adversarial(E, k):
    """N runs, IUPAC letters, '-', lowercase, records of length k-1 / k / k+1, clean runs of exactly k at the
    record end (split_kmer.rs:89 quirk), homopolymers (palindromes when rc) and tile-boundary straddlers."""
    rng = np.random.default_rng(7 * k)
    base = rand_records(rng, 4, 9000, p_bad=0.01, lower=True)
    core = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=k + 1).tolist())
    edge = [core[:k - 1], core[:k], core[:k + 1], b"N" + core[:k], core[:k] + b"N", core[:k] + b"N" + core[:k],
            core[:k + 1] + b"N" + core[:k + 1], b"A" * (3 * k), b"AT" * (2 * k), b"ACGT" * k, b"", b"N" * 50,
            b"acgtn" * 20, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=4096 - k // 2).tolist()),
            bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=8200).tolist())]
    for rc in (True, False):
        check_dicts(E, [base + edge, edge + base, [core[:k + 1]]], k, rc)


@pytest.mark.parametrize("k", [31, 41])
def test_dict_larger_genome(E, k):
    """~300 kbp related samples: many tiles and buckets, exercises cross-tile halos."""
    import synth
    anc = synth.ancestor(300_000, seed=3)
    streams = [synth.sample_stream(anc, i, 4, private_snps=40, shared_snps=10, seed=3) for i in range(4)]
    ds = E.DictSet.build([s.tobytes() for s in streams], k, True)
    for i, s in enumerate(streams):
        recs = s.tobytes().split(b"\n")[:-1]
        ok, ob = oracle_dict(recs, k, True).export()
        gk, gb = ds.export(i)
        assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)


def as_map(keys, var, counts):
    return {(int(k["hi"]) << 64) | int(k["lo"]): (bytes(v), int(c)) for k, v, c in zip(keys, var, counts)}


def build_both(E, samples, k, rc):
    names = [f"s{i}" for i in range(len(samples))]
    ds = E.DictSet.build([E.record_stream(r) for r in samples], k, rc)
    ga = ds.merge(names)
    oa = ora.Array.from_dicts([oracle_dict(r, k, rc) for r in samples], names)
    return ga, oa


@pytest.mark.parametrize("k,rc", [(9, True), (15, False), (31, True), (33, True), (41, False), (63, True)])
def test_merge_array(E, k, rc):
    rng = np.random.default_rng(k)
    anc = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=5000).tolist())
    samples = []
    for i in range(7):
        s = bytearray(anc)
        for p in rng.integers(0, len(s), size=25):
            s[p] = b"ACGT"[rng.integers(0, 4)]
        samples.append([bytes(s[:2500]), bytes(s[2500:])] + rand_records(rng, 1, 300))
    ga, oa = build_both(E, samples, k, rc)
    assert ga.names == oa.names and ga.nkmers == oa.nkmers
    assert as_map(*ga.export()) == as_map(*oa.export())
    assert list(ga.sample_kmers()) == [int(x) for x in (oa.export()[1] != ord("-")).sum(axis=0)]


@pytest.mark.parametrize("k,rc", [(31, True), (41, True), (63, False)])
def test_merge_array_many_samples(E, k, rc):
    """More samples than a wave deals with at once (70 > 64), several sub-buckets per bucket and slices longer than one batch of
    look-ups: the union / assemble kernels' sample loops, batched probes and insert paths, 64- and 128-bit, against the oracle."""
    rng = np.random.default_rng(1000 + k)
    anc = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=120_000)
    samples = []
    for i in range(70):
        s = anc.copy()
        pos = rng.integers(0, len(s), size=40)
        s[pos] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=40)
        recs = [bytes(s[:70_000]), bytes(s[70_000:])]
        if i % 7 == 3:
            recs.append(bytes(s[1000:1400]))                       # a repeat: ambiguity codes where the copies differ from nothing, counts unchanged
        samples.append(recs)
    ga, oa = build_both(E, samples, k, rc)
    assert ga.names == oa.names and ga.nkmers == oa.nkmers
    assert as_map(*ga.export()) == as_map(*oa.export())
    assert list(ga.sample_kmers()) == [int(x) for x in (oa.export()[1] != ord("-")).sum(axis=0)]


@pytest.mark.parametrize("k", [31, 21])
def test_a_sample_that_is_one_repeat_takes_the_sorted_path(E, k, capfd, monkeypatch):
    """A sample that is one repeat puts all its words into one region: the extraction kernel's fixed-capacity layout overflows, the batch is
    extracted again with exact offsets and its dictionaries are sorted, and the merge must come out of the sorted path unharmed (same array
    as the oracle; ambiguity codes where copies differ).  A batch of ordinary samples goes through the append pass (SKX_DEBUG tells which).
    (Until round 5 the append pass had a way out of its own for such samples -- a wave's queue of kept words could overflow; the queue now
    holds a batch's leftover plus a whole load and cannot.)"""
    rng = np.random.default_rng(40 + k)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    anc = acgt[rng.integers(0, 4, size=60_000)]
    ordinary = []
    for i in range(20):
        s = anc.copy()
        pos = rng.integers(0, len(s), size=30)
        s[pos] = acgt[rng.integers(0, 4, size=30)]
        ordinary.append([bytes(s[:30_000]), bytes(s[30_000:])])
    monkeypatch.setenv("SKX_DEBUG", "1")
    ga, oa = build_both(E, ordinary, k, True)
    assert as_map(*ga.export()) == as_map(*oa.export())
    assert "append: logQ=" in capfd.readouterr().err and True
    unit = acgt[rng.integers(0, 4, size=70)].tobytes()
    odd = ordinary[:6] + [[b"A" * 120_000, unit * 1500], [b"AC" * 50_000 + bytes(anc[:5000])]]
    ga, oa = build_both(E, odd, k, True)
    assert "sorted: the dictionaries are sorted" in capfd.readouterr().err
    assert ga.names == oa.names and ga.nkmers == oa.nkmers
    assert as_map(*ga.export()) == as_map(*oa.export())
    assert list(ga.sample_kmers()) == [int(x) for x in (oa.export()[1] != ord("-")).sum(axis=0)]


FILTERS = [(ft, amb, mask, gaps) for ft in range(4) for amb in (False, True) for mask in (False, True) for gaps in (False, True)]


@pytest.mark.parametrize("min_freq", [0.0, 0.5, 0.9, 1.0])
def test_filter_and_align(E, min_freq):
    rng = np.random.default_rng(5)
    anc = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3000).tolist())
    samples = []
    for i in range(6):
        s = bytearray(anc)
        for p in rng.integers(0, len(s), size=30):
            s[p] = b"ACGT"[rng.integers(0, 4)]
        if i % 2:
            s = s[:2000]           # missing k-mers
        samples.append([bytes(s), bytes(s[100:400])])   # repeats -> ambiguity codes
    for ft, amb, mask, gaps in FILTERS:
        ga, oa = build_both(E, samples, 9, True)
        g = ga.align(filter_type=ft, mask_ambig=mask, ignore_const_gaps=gaps, min_freq=min_freq, filter_ambig_as_missing=amb)
        o = oa.align(filter_type=ft, mask_ambig=mask, ignore_const_gaps=gaps, min_freq=min_freq, filter_ambig_as_missing=amb)
        assert G.aln_length(g) == G.aln_length(o), (ft, amb, mask, gaps)
        assert sorted(zip(*[l for l in g.decode().splitlines()[1::2]])) == sorted(zip(*[l for l in o.decode().splitlines()[1::2]]))
    # removed-row counts of MergeSkaArray::filter
    for ft, amb, mask, gaps in FILTERS[::3]:
        ga, oa = build_both(E, samples, 9, True)
        assert ga.filter(3, amb, ft, mask, gaps, True) == oa.filter(3, amb, ft, mask, gaps, True)
        assert as_map(*ga.export()) == as_map(*oa.export())


@pytest.mark.parametrize("filt_ambig", [True, False])
def test_distance(E, filt_ambig):
    rng = np.random.default_rng(11)
    anc = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=4000).tolist())
    samples = []
    for i in range(9):
        s = bytearray(anc)
        for p in rng.integers(0, len(s), size=40):
            s[p] = b"ACGT"[rng.integers(0, 4)]
        samples.append([bytes(s[: 4000 - 150 * i]), bytes(s[50:500])])
    ga, oa = build_both(E, samples, 9, True)
    for mf in (0.0, 0.6):
        ga, oa = build_both(E, samples, 9, True)
        assert ga.distance_tsv(min_freq=mf, filt_ambig=filt_ambig) == oa.distance_tsv(min_freq=mf, filt_ambig=filt_ambig)


def test_skf_written_by_engine_is_read_by_oracle(E, tmp_path):
    a = E.Array.build(G.fasta_inputs(E, [G.fin("test_1.fa"), G.fin("test_2.fa")]), k=17)
    p = str(tmp_path / "e.skf")
    a.save(p)
    o = ora.Array.load(p)
    ref = ora.Array.load(G.fin("merge.skf"))
    assert as_map(*o.export()) == as_map(*ref.export()) and o.names == ref.names


@pytest.mark.parametrize("k", [15, 41])
def test_keyset_exchange_equals_single_merge(E, k):
    """Section 8e: union of per-shard key tables, then per-shard assemble == one merge of everything."""
    rng = np.random.default_rng(2)
    anc = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=6000).tolist())
    samples = []
    for i in range(6):
        s = bytearray(anc)
        for p in rng.integers(0, len(s), size=20):
            s[p] = b"ACGT"[rng.integers(0, 4)]
        samples.append([bytes(s)])
    names = [f"s{i}" for i in range(6)]
    whole = E.DictSet.build([E.record_stream(r) for r in samples], k, True).merge(names)
    shards = [E.DictSet.build([E.record_stream(r) for r in samples[a:b]], k, True) for a, b in ((0, 2), (2, 6))]
    keysets = [s.union_keys() for s in shards]
    rows = E.KeySet.merge(keysets)
    assert len(rows) == whole.nkmers
    wk, wv, _ = whole.export()
    col = 0
    for s, (a, b) in zip(shards, ((0, 2), (2, 6))):
        part = s.assemble(rows, names[a:b])
        pk, pv, _ = part.export()
        assert np.array_equal(pk["lo"], wk["lo"]) and np.array_equal(pk["hi"], wk["hi"]) and np.array_equal(pv, wv[:, a:b])
        col += b - a


def _write_fastq(path, reads, quals):
    with open(path, "wb") as f:
        for i, (r, q) in enumerate(zip(reads, quals)):
            f.write(b"@r%d\n" % i + r + b"\n+\n" + q + b"\n")


@pytest.fixture(scope="module")
def fastq_pair(tmp_path_factory):
    """~10 M read bases with 2 % substitution errors: millions of distinct k-mers, so the blocked bloom filter of
    bloom_filter.rs produces false positives and the order-dependent part of KmerFilter is exercised."""
    rng = np.random.default_rng(42)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300_000)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    n_reads, L = 34_000, 150
    d = tmp_path_factory.mktemp("fq")
    paths = []
    for tag in ("fwd", "rev"):
        starts = rng.integers(0, len(genome) - L, size=n_reads)
        reads, quals = [], []
        for s in starts:
            r = genome[s:s + L].copy()
            if tag == "rev":
                r = comp[r[::-1]]
            err = rng.random(L) < 0.02
            r[err] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(err.sum()))
            if rng.random() < 0.01:
                r[rng.integers(0, L)] = ord("N")
            q = rng.choice(np.frombuffer(b"#+5?I", dtype=np.uint8), size=L, p=[0.02, 0.03, 0.05, 0.2, 0.7])
            reads.append(r.tobytes())
            quals.append(q.tobytes())
        p = str(d / f"s_{tag}.fastq")
        _write_fastq(p, reads, quals)
        paths.append(p)
    return paths


@pytest.mark.parametrize("k,min_count,qf,min_qual", [(31, 1, 2, 20), (31, 2, 2, 20), (31, 3, 2, 20), (31, 5, 0, 20), (21, 4, 1, 10),
                                                     (41, 2, 2, 20), (41, 3, 1, 5)])
def test_fastq_kmer_filter(E, fastq_pair, k, min_count, qf, min_qual):
    f1, f2 = fastq_pair
    og = ora.Dict.from_files(k, f1, f2, True, ora.qual(min_count, min_qual, qf))
    ds = E.DictSet.from_files([(f1, f2)], k, True, E.qual(min_count, min_qual, qf))
    ok, ob = og.export()
    gk, gb = ds.export(0)
    assert len(gk) == len(ok) and len(ok) > 1000
    assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)


@pytest.fixture(scope="module")
def ragged_fastq(tmp_path_factory):
    """Reads of every awkward length -- shorter than k, exactly k, k + 1, odd, long -- with N runs, lower-case bases and the whole quality range;
    total length no multiple of 16, deep enough (a 3 kbp genome) that every count threshold is met by some k-mers and missed by others."""
    rng = np.random.default_rng(7)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3_000)
    reads, quals = [], []
    for i in range(6_000):
        L = int(rng.choice([3, 6, 7, 8, 14, 15, 16, 17, 18, 31, 32, 33, 41, 42, 63, 64, 65, 97, 151]))
        s = int(rng.integers(0, len(genome) - L))
        r = genome[s:s + L].copy()
        err = rng.random(L) < 0.01
        r[err] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(err.sum()))
        if rng.random() < 0.05:
            a = int(rng.integers(0, L)); r[a:a + int(rng.integers(1, 4))] = ord("N")
        if rng.random() < 0.05:
            r = np.frombuffer(r.tobytes().lower(), dtype=np.uint8).copy()
        q = rng.integers(100, 127, size=L).astype(np.uint8)          # mostly high, one in ten anywhere in '!' .. '~'
        lowq = rng.random(L) < 0.1
        q[lowq] = rng.integers(33, 127, size=int(lowq.sum())).astype(np.uint8)
        reads.append(r.tobytes()); quals.append(q.tobytes())
    d = tmp_path_factory.mktemp("rfq")
    p = str(d / "ragged.fastq")
    _write_fastq(p, reads, quals)
    return p


@pytest.mark.parametrize("k,rc,min_count,qf,min_qual", [(7, True, 2, 2, 20), (15, True, 3, 1, 40), (17, False, 2, 2, 0), (31, True, 3, 2, 60),
                                                        (33, False, 4, 0, 20), (63, True, 2, 2, 20), (63, False, 3, 1, 80)])
def test_fastq_kmer_filter_ragged_reads(E, ragged_fastq, k, rc, min_count, qf, min_qual):
    """The window pass at its edges: k below 16 (the base that leaves a window lies in the thread's own sixteen positions), k = 63 (the whole 64-position
    history), single strand, reads shorter than k, every quality threshold from 0 to the top of the range; one file (ska_dict.rs:118-180)."""
    og = ora.Dict.from_files(k, ragged_fastq, None, rc, ora.qual(min_count, min_qual, qf))
    ds = E.DictSet.from_files([(ragged_fastq, None)], k, rc, E.qual(min_count, min_qual, qf))
    ok, ob = og.export()
    gk, gb = ds.export(0)
    assert len(ok) > 50
    assert len(gk) == len(ok) and np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)


@pytest.mark.parametrize("layout", [2, 3])
def test_fastq_kmer_filter_in_every_partition_layout(E, fastq_pair, layout, monkeypatch):
    """The count filter's partition passes have three instantiations -- 65 536 partitions of 48 bloom words, 131 072 of 24 (k = 17 on a deep isolate),
    262 144 of 12 (beyond ~270 M windows) -- chosen by the sample's size; SKX_KNOBS=reads_layout forces the other two on this small sample."""
    f1, f2 = fastq_pair
    k, min_count = 31, 3
    og = ora.Dict.from_files(k, f1, f2, True, ora.qual(min_count, 20, 2))
    ok, ob = og.export()
    monkeypatch.setenv("SKX_KNOBS", f"reads_layout={layout}")
    ds = E.DictSet.from_files([(f1, f2)], k, True, E.qual(min_count, 20, 2))
    gk, gb = ds.export(0)
    assert len(gk) == len(ok) and np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)


# ---- failure behaviour (the reference panics; the C ABI returns codes with the same message text) ----
def test_error_codes(E, tmp_path):
    with pytest.raises(E.EngineError) as ei:                                   # ska_dict.rs:342-344
        E.DictSet.build([b"ACGTACGTACGT\n"], 6, True)
    assert ei.value.code == E.EINVAL and "Invalid k-mer length" in str(ei.value)
    with pytest.raises(E.EngineError) as ei:
        E.DictSet.build([b"ACGTACGTACGT\n"], 65, True)
    assert ei.value.code == E.EINVAL
    with pytest.raises(E.EngineError) as ei:                                   # ska_dict.rs:374-376
        E.DictSet.build([b"ACGTACGTACGTACGTACGTACGTACGTACGTAAA\n", b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n"], 31, True)
    assert ei.value.code == E.EEMPTY and "no valid sequence" in str(ei.value)
    p = str(tmp_path / "empty.fa")
    open(p, "w").close()
    with pytest.raises(E.EngineError) as ei:                                   # ska_dict.rs:131-132
        E.Array.build([("e", p, None)], k=15)
    assert ei.value.code == E.EIO and "Invalid path/file" in str(ei.value)
    with pytest.raises(E.EngineError):
        E.Array.build([("e", str(tmp_path / "missing.fa"), None)], k=15)
    with pytest.raises(E.EngineError) as ei:                                   # not a .skf
        E.Array.load(G.fin("test_1.fa"))
    assert ei.value.code == E.EFORMAT
    # merge_ska_dict.rs:78-87: k / strand mismatches between the two halves of a merge
    a = E.DictSet.build([b"ACGTTGCAAGGCTTAACCGGTTAAGC\n"], 9, True)
    b = E.DictSet.build([b"ACGTTGCAAGGCTTAACCGGTTAAGC\n"], 11, True)
    with pytest.raises(E.EngineError) as ei:
        a.assemble(b.union_keys(), ["x"])
    assert "K-mer lengths do not match" in str(ei.value)
    c = E.DictSet.build([b"ACGTTGCAAGGCTTAACCGGTTAAGC\n"], 9, False)
    with pytest.raises(E.EngineError) as ei:
        a.assemble(c.union_keys(), ["x"])
    assert "Strand use inconsistent" in str(ei.value)


def test_ragged_and_tiny_inputs(E):
    """empty records, a sample that is one k+1 record, single-sample arrays, k=5 (8-bit keys)."""
    samples = [[b"", b"ACGTAC", b"", b"ACGTACGT" * 3], [b"ACGTAC"], [b"TTTTTTGACCA", b""]]
    check_dicts(E, samples, 5, True)
    check_dicts(E, [samples[0], samples[2]], 7, False)
    ga, oa = build_both(E, [[b"ACGTACGTTGCA" * 4]], 9, True)
    assert as_map(*ga.export()) == as_map(*oa.export())
    g = ga.align(filter_type=E.FILTER_NONE, min_freq=0.0).decode().splitlines()
    o = oa.align(filter_type=ora.FILTER_NONE, min_freq=0.0).decode().splitlines()
    assert g[0] == o[0] and sorted(g[1]) == sorted(o[1])
    assert ga.nk(True).decode().split("\n")[0].startswith("ska_version=")


def test_repeat_rich_tiny_wide_sample(E):
    """k > 31 with fewer buckets than staging rounds and one bucket holding most of a tile (poly-A + a short random tail):
    the words of that bucket go out through the unstaged path of the 128-bit scatter kernel."""
    rng = np.random.default_rng(17)
    tail = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=600).tolist())
    for k in (33, 41, 63):
        check_dicts(E, [[b"A" * 3300 + tail], [tail + b"AT" * 1500]], k, True)
        check_dicts(E, [[b"C" * 3900]], k, False)


def test_repeat_rich_sample(E):
    """Tandem repeats / homopolymers put tens of thousands of identical split k-mers into one hash bucket: the region
    overflows the fixed-capacity layout (-> exact histogram pass) and the counting sort (-> table dedupe)."""
    rng = np.random.default_rng(9)
    rnd = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=150_000).tolist())
    recs = [rnd[:60_000] + b"AT" * 12_000 + rnd[60_000:90_000], b"A" * 20_000 + rnd[90_000:], b"GATTACA" * 5_000, b"ACGT" * 9_000]
    for k, rc in ((31, True), (15, False)):
        ds = check_dicts(E, [recs, [rnd]], k, rc)
        ga = ds.merge(["rep", "rnd"])
        oa = ora.Array.from_dicts([oracle_dict(recs, k, rc), oracle_dict([rnd], k, rc)], ["rep", "rnd"])
        assert as_map(*ga.export()) == as_map(*oa.export())


def test_unrelated_samples_fine_split(E):
    """Unrelated genomes: the union is ~32x one sample, so the row keyset is split much finer (2^6 sub-buckets per
    dictionary bucket) than the 16-entry sub-index the dedupe kernel leaves per region -> bracketed slice search."""
    rng = np.random.default_rng(77)
    samples = [[bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=100_000).tolist())] for _ in range(32)]
    ga, oa = build_both(E, samples, 31, True)
    assert ga.nkmers == oa.nkmers
    gk, gv, gc = ga.export()
    ok, ov, oc = oa.export()
    go, oo = np.argsort(gk["lo"], kind="stable"), np.argsort(ok["lo"], kind="stable")
    assert np.array_equal(gk["lo"][go], ok["lo"][oo])
    assert np.array_equal(np.asarray(gv)[go], np.asarray(ov)[oo])
    assert np.array_equal(np.asarray(gc)[go], np.asarray(oc)[oo])


def _related_samples(rng, n, length=4000, snps=30):
    anc = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length).tolist())
    out = []
    for _ in range(n):
        s = bytearray(anc)
        for p in rng.integers(0, length, size=snps):
            s[p] = b"ACGT"[rng.integers(0, 4)]
        out.append([bytes(s[: length // 2]), bytes(s[length // 2:])])
    return bytes(anc), out


@pytest.mark.parametrize("k,rc", [(15, True), (31, True), (31, False), (41, True)])
def test_skf_lifecycle_merge_delete_weed(E, k, rc, tmp_path):
    """`ska merge` / `ska delete` / `ska weed` on the device vs the oracle's restatement (generic_modes.rs:90-106,192-267),
    for arrays in engine order (just built) and in file order (loaded), 64- and 128-bit keys."""
    rng = np.random.default_rng(500 + k)
    anc, samples = _related_samples(rng, 5)
    names = [f"s{i}" for i in range(5)]

    def both(idx):
        ds = E.DictSet.build([E.record_stream(samples[i]) for i in idx], k, rc)
        ga = ds.merge([names[i] for i in idx])
        oa = ora.Array.from_dicts([oracle_dict(samples[i], k, rc) for i in idx], [names[i] for i in idx])
        return ga, oa

    g1, o1 = both([0, 1, 2])
    g2, o2 = both([3, 4])
    for via_file in (False, True):
        a1, a2 = g1, g2
        if via_file:
            p1, p2 = str(tmp_path / "a1.skf"), str(tmp_path / "a2.skf")
            g1.save(p1), g2.save(p2)
            a1, a2 = E.Array.load(p1), E.Array.load(p2)
        gm, om = E.Array.merge([a1, a2]), ora.Array.merge([o1, o2])
        assert gm.names == om.names == names and gm.nkmers == om.nkmers
        assert as_map(*gm.export()) == as_map(*om.export())
        assert list(gm.sample_kmers()) == [int(x) for x in (om.export()[1] != ord("-")).sum(axis=0)]
        # merging is building everything together
        gall, oall = both([0, 1, 2, 3, 4])
        assert as_map(*gm.export()) == as_map(*gall.export())
        # delete two samples (one from each input); rows only they had disappear
        gm.delete_samples(["s1", "s3"]), om.delete_samples(["s1", "s3"])
        assert gm.names == om.names == ["s0", "s2", "s4"]
        assert as_map(*gm.export()) == as_map(*om.export())
        # weed: a stretch of the ancestor plus something foreign, both directions, with and without the follow-up filter
        wf = str(tmp_path / "weed.fa")
        with open(wf, "wb") as f:
            f.write(b">w1\n" + anc[500:1500] + b"\n>w2\n" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300).tolist()) + b"\n")
        for reverse, kw in ((False, {}), (True, dict(min_freq=1.0, filter_type=E.FILTER_NO_CONST)), (False, dict(ambig_mask=True, min_freq=0.0))):
            gw, ow = E.Array.merge([a1, a2]), ora.Array.merge([o1, o2])
            okw = {("filter_type" if kk == "filter_type" else kk): (vv if kk != "filter_type" else ora.FILTER_NO_CONST) for kk, vv in kw.items()}
            gw.weed(wf, reverse=reverse, **kw), ow.weed(wf, reverse=reverse, **okw)
            assert gw.nkmers == ow.nkmers and 0 < gw.nkmers
            assert as_map(*gw.export()) == as_map(*ow.export())
    # weed with a key set taken straight from a FASTA (RefSka::new): count of removed rows
    ks = E.KeySet.from_fasta(wf, k, rc)
    gw = E.Array.merge([g1, g2])
    before = gw.nkmers
    removed = gw.weed_keys(ks)
    assert removed == before - gw.nkmers and removed > 0


def test_skf_lifecycle_errors(E, tmp_path):
    rng = np.random.default_rng(3)
    _, samples = _related_samples(rng, 2, length=600, snps=5)
    a = E.DictSet.build([E.record_stream(samples[0])], 31, True).merge(["a"])
    b17 = E.DictSet.build([E.record_stream(samples[1])], 17, True).merge(["b"])
    bss = E.DictSet.build([E.record_stream(samples[1])], 31, False).merge(["b"])
    b = E.DictSet.build([E.record_stream(samples[1])], 31, True).merge(["b"])
    with pytest.raises(E.EngineError, match="K-mer lengths do not match"):
        E.Array.merge([a, b17])
    with pytest.raises(E.EngineError, match="Strand use inconsistent"):
        E.Array.merge([a, bss])
    m = E.Array.merge([a, b])
    with pytest.raises(E.EngineError, match="Invalid number of samples to remove"):
        m.delete_samples([])
    with pytest.raises(E.EngineError, match="Invalid number of samples to remove"):
        m.delete_samples(["a", "b"])
    with pytest.raises(E.EngineError, match="Could not find sample"):
        m.delete_samples(["zzz"])
    with pytest.raises(E.EngineError, match="Cannot create reference from FASTQ"):
        m.weed(G.fin("test_1_fwd.fastq.gz"))
    # filtered for output only (update_kmers = false): keys and rows are out of step, set operations refuse
    m.apply_filters(0.0, filter_type=E.FILTER_NO_CONST)
    if m.nkmers != m.nrows:
        with pytest.raises(E.EngineError, match="out of step"):
            m.delete_samples(["a"])


SKF_MODES = {"host": {"skf_device": "0"}, "device": {"skf_device": "1"},
             "device_small_groups": {"skf_device": "1", "skf_group_chunks": "3"}}


@pytest.mark.parametrize("mode", list(SKF_MODES))
def test_skf_stream_codec_integrity(E, tmp_path, mode, monkeypatch):
    """The streaming codec: a multi-super-block file round-trips (engine -> engine, engine -> oracle, oracle -> engine),
    and a flipped byte / a truncated file is an error (masked CRC-32C per chunk), never a silently different array.
    Once with the `variants` section on the host thread team, once on the device (snappy + CRC-32C + CBOR cells in
    skx_snappy.hip), once on the device in groups of 3 chunks (rows straddle groups)."""
    set_knob(monkeypatch, "skf_block_mb", "4")       # read once, at the codec's first use: many super-blocks in this test
    for kk, vv in SKF_MODES[mode].items():
        set_knob(monkeypatch, kk, vv)
    rng = np.random.default_rng(11)
    _, samples = _related_samples(rng, 40, length=60_000, snps=300)
    names = [f"s{i}" for i in range(40)]
    ga = E.DictSet.build([E.record_stream(s) for s in samples], 31, True).merge(names)      # ~0.4 M rows x 40: 33 MB of CBOR
    p = str(tmp_path / "big.skf")
    ga.save(p)
    ref = as_map(*ga.export())
    assert as_map(*E.Array.load(p).export()) == ref
    set_knob(monkeypatch, "skf_device", "1" if mode == "host" else "0")                   # written one way, read the other
    assert as_map(*E.Array.load(p).export()) == ref
    set_knob(monkeypatch, "skf_device", SKF_MODES[mode]["skf_device"])
    oa = ora.Array.load(p)
    assert oa.names == names and as_map(*oa.export()) == ref
    p2 = str(tmp_path / "big_oracle.skf")
    oa.save(p2)                                                                              # uncompressed chunks (type 0x01)
    assert as_map(*E.Array.load(p2).export()) == ref
    raw = bytearray(open(p, "rb").read())
    raised = 0
    for frac in (0.31, 0.5, 0.62, 0.77, 0.93):                  # a flipped bit is an error -- or, where it only redirects a copy to
        bad = bytearray(raw)                                    # identical bytes, the same array; never a silently different one
        bad[int(len(bad) * frac)] ^= 0x40
        open(str(tmp_path / "flip.skf"), "wb").write(bad)
        try:
            got = E.Array.load(str(tmp_path / "flip.skf"), want_bits=64)
        except E.EngineError:
            raised += 1
            continue
        assert as_map(*got.export()) == ref
    assert raised >= 3
    open(str(tmp_path / "cut.skf"), "wb").write(raw[: len(raw) * 2 // 3])
    with pytest.raises(E.EngineError):
        E.Array.load(str(tmp_path / "cut.skf"), want_bits=64)
    open(str(tmp_path / "junk.skf"), "wb").write(b"not a snappy stream at all")
    with pytest.raises(E.EngineError):
        E.Array.load(str(tmp_path / "junk.skf"), want_bits=64)


@pytest.mark.parametrize("k,rc", [(15, True), (31, True), (31, False), (41, True)])
def test_map_vs_oracle(E, k, rc, tmp_path):
    """`ska map` (RefSka::new + map + AlnWriter + VCF, ska_ref.rs) on the device vs the oracle: multi-chromosome reference with
    N runs, lower case, a repeated block and a contig shorter than k; arrays in engine order and in file order; all flag
    combinations; text compared byte for byte."""
    rng = np.random.default_rng(900 + k)
    anc, samples = _related_samples(rng, 4, length=6000, snps=40)
    names = [f"s{i}" for i in range(4)]
    ga, oa = build_both(E, samples, k, rc)
    ref = bytearray(anc)
    ref[700:720] = b"N" * 20
    ref[1500:1600] = bytes(ref[1500:1600]).lower()
    rep = bytes(ref[2000:2200])
    chroms = [bytes(ref[:3000]), b"ACGTACG", bytes(ref[3000:]) + rep, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=500).tolist())]
    rp = str(tmp_path / "ref.fa")
    with open(rp, "wb") as f:
        for i, c in enumerate(chroms):
            f.write(b">chr%d some description\n" % i)
            for o in range(0, len(c), 70):
                f.write(c[o:o + 70] + b"\n")
    p = str(tmp_path / "a.skf")
    ga.save(p)
    for arr in (ga, E.Array.load(p)):
        for fmt in ("aln", "vcf"):
            for ambig_mask in (False, True):
                for repeat_mask in (False, True):
                    g = arr.map(rp, fmt=fmt, ambig_mask=ambig_mask, repeat_mask=repeat_mask)
                    o = oa.map(rp, fmt=fmt, ambig_mask=ambig_mask, repeat_mask=repeat_mask)
                    assert g == o, (fmt, ambig_mask, repeat_mask)
    with pytest.raises(E.EngineError, match="Cannot create reference from FASTQ"):
        ga.map(G.fin("test_1_fwd.fastq.gz"))
    far = str(tmp_path / "far.fa")
    open(far, "wb").write(b">x\n" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=400).tolist()) + b"\n")
    with pytest.raises(E.EngineError, match="No split k-mers mapped"):
        ga.map(far)
    tiny = str(tmp_path / "tiny.fa")
    open(tiny, "wb").write(b">x\nACGT\n")
    with pytest.raises(E.EngineError, match="has no valid sequence"):
        ga.map(tiny)


@pytest.mark.parametrize("k,rc", [(9, True), (21, False), (31, True), (33, True), (63, False)])
def test_cov_histogram_and_fit(E, fastq_pair, k, rc):
    """`ska cov` (coverage.rs): the device's occurrence-count histogram is the oracle's hash-map count, bin for bin; the host
    fit gives the oracle's cutoff and parameters; plot_hist's table agrees (densities to 1e-12 relative: two compilers)."""
    f1, f2 = fastq_pair[0], fastq_pair[1]
    gh, oh = E.cov_histogram(f1, f2, k, rc), ora.cov_histogram(f1, f2, k, rc)
    assert np.array_equal(gh, oh) and gh.sum() > 0
    for files in ((f1, f2), (G.fin("test_long_1_fwd.fastq.gz"), G.fin("test_long_1_rev.fastq.gz"))):
        try:
            ot, oc = ora.cov(*files, k=k, rc=rc)
        except ora.OracleError:                       # "Optimiser did not converge" is a legitimate outcome (the reference panics)
            with pytest.raises(E.EngineError, match="did not converge"):
                E.cov(*files, k=k, rc=rc)
            continue
        gt, gc = E.cov(*files, k=k, rc=rc)
        assert gc == oc
        gl, ol = gt.decode().splitlines(), ot.decode().splitlines()
        assert len(gl) == len(ol) and gl[0] == ol[0]
        for a, b in zip(gl[1:], ol[1:]):
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[:2] == fb[:2] and fa[3] == fb[3] and abs(float(fa[2]) - float(fb[2])) <= 1e-12 * abs(float(fb[2]))
    w0, c, cut = E.cov_fit(G.COV_EXAMPLE)
    ow0, oc_, ocut = ora.cov_fit(G.COV_EXAMPLE)
    assert cut == ocut == 9 and abs(w0 - ow0) < 1e-9 and abs(c - oc_) < 1e-7
    with pytest.raises(E.EngineError, match="appears to be FASTA"):
        E.cov(G.fin("test_1.fa"), G.fin("test_2.fa"), k=9)


@pytest.mark.parametrize("k,rc,batch_mb", [(31, True, 1), (31, False, 20), (41, True, 20), (15, True, 30)])
def test_build_in_batches_equals_one_batch(E, k, rc, batch_mb, tmp_path, monkeypatch, capfd):
    """`ska build` on more samples than the device-memory budget admits at once: batches of samples are built and joined by
    the `ska merge` row-set path; rows, columns, names and counts must equal the single-batch build and the oracle's."""
    rng = np.random.default_rng(900 + k + batch_mb)
    _, samples = _related_samples(rng, 9, length=6000, snps=40)
    samples[4] = samples[4] + rand_records(rng, 2, 500)
    inputs = []
    for i, recs in enumerate(samples):
        p = tmp_path / f"b{i}.fa"
        p.write_bytes(b"".join(b">r%d\n%s\n" % (j, r) for j, r in enumerate(recs)))
        inputs.append((f"s{i}", str(p), None))
    monkeypatch.delenv("SKX_BUILD_BATCH_MB", raising=False)
    one = E.Array.build(inputs, k=k, rc=rc, threads=2)
    monkeypatch.setenv("SKX_BUILD_BATCH_MB", str(batch_mb))        # 8 MB fixed + 24 B/base per sample: 1 -> one sample per batch
    monkeypatch.setenv("SKX_DEBUG", "1")
    capfd.readouterr()
    many = E.Array.build(inputs, k=k, rc=rc, threads=2)
    n_batches = capfd.readouterr().err.count("as one batch")
    assert n_batches == {1: 9, 20: 5, 30: 3}[batch_mb]
    monkeypatch.delenv("SKX_BUILD_BATCH_MB")
    monkeypatch.delenv("SKX_DEBUG")
    oa = ora.Array.from_dicts([oracle_dict(r, k, rc) for r in samples], [x[0] for x in inputs])
    assert many.names == one.names == oa.names and many.nkmers == one.nkmers == oa.nkmers
    m = as_map(*many.export())
    assert m == as_map(*one.export()) and m == as_map(*oa.export())
    assert list(many.sample_kmers()) == list(one.sample_kmers())
    # the joined array goes on through filter + align like any other
    f1, f2, f3 = (a.align(filter_type=1, min_freq=0.8) for a in (many, one, oa))
    cols = lambda t: sorted(zip(*[l for l in t.decode().splitlines()[1::2]]))
    assert cols(f1) == cols(f2) == cols(f3)


def _sorted_export(arr):
    keys, var, counts = arr.export()
    order = np.lexsort((keys["lo"], keys["hi"]))
    return keys["hi"][order], keys["lo"][order], np.asarray(var)[order], np.asarray(counts)[order]


@pytest.mark.parametrize("k,length,contigs", [(31, 41_000_000, 2), (41, 34_000_000, 2), (31, 100_000_000, 1), (31, 12_000_000, 3)])
def test_large_assembly_stays_on_the_assembly_kernels(E, k, length, contigs, monkeypatch):
    """The reference's add_file_kmers has no size limit (ska_dict.rs:118-180).  Until round 6 a sample's regions were capped at what the
    per-region LDS sort holds, so the bucket count grew with the genome (17x the time per base at 40 Mbp) and beyond 2^13 regions -- 40.1 Mbp at
    k <= 31, 33.5 Mbp above -- the sample left the assembly kernels for the read sets' sort-based form.  Now the regions grow instead: such a sample
    is extracted by the same kernel into 2^10 (2^11) regions, merged by the append pass (skx_ctx_merge_path says so), and its sorted dictionary --
    what skx_dictset_export asks for -- comes from the flat sort.  The merged array with an ordinary sample and both dictionaries equal the
    oracle's; SKX_KNOBS=sorted_dicts sends the same samples through the sorted path: the same array."""
    rng = np.random.default_rng(k + contigs)
    g = rng.integers(0, 4, size=length, dtype=np.uint8)
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[g]
    del g
    small = big[: 1_000_000].copy()
    small[rng.integers(0, len(small), size=300)] = ord("A")
    cuts = [length * i // contigs for i in range(contigs + 1)]
    samples = [[big[a:b].tobytes() for a, b in zip(cuts[:-1], cuts[1:])], [small.tobytes()]]
    streams = [E.record_stream(r) for r in samples]
    ods = [oracle_dict(r, k, True) for r in samples]
    oa = ora.Array.from_dicts(ods, ["big", "small"])
    want = _sorted_export(oa)
    ds = E.DictSet.build(streams, k, True)
    ga = ds.merge(["big", "small"])                                   # the dictionaries as extracted: the append pass
    assert E.default_context().merge_path() == ("append128" if k > 31 else "append64")
    assert ga.nkmers == oa.nkmers
    for x, y in zip(_sorted_export(ga), want):
        assert np.array_equal(x, y)
    del ga
    for i in range(2):                                                # a sample's SkaDict as such: the flat sort
        ok, ob = ods[i].export()
        gk, gb = ds.export(i)
        assert len(gk) == len(ok)
        assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)
    if length > 50_000_000:
        return                 # (the sorted path's half once at each key width is enough for the suite's running time)
    ga = ds.merge(["big", "small"])                                   # the same dictset, sorted by now: union + assemble
    assert E.default_context().merge_path().startswith("sorted: the dictionaries are sorted")
    for x, y in zip(_sorted_export(ga), want):
        assert np.array_equal(x, y)
    ds.free()
    if length > 15_000_000:
        return                 # (the rebuild under the knob once, on the 12 Mbp case)
    set_knob(monkeypatch, "sorted_dicts", "1")
    gs = E.DictSet.build(streams, k, True).merge(["big", "small"])
    assert E.default_context().merge_path().startswith("sorted:")
    for x, y in zip(_sorted_export(gs), want):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n_samples,pad", [(1, ""), (2, "x"), (3, ""), (7, "yy"), (33, "z")])
def test_skf_device_codec_alignments(E, tmp_path, n_samples, pad, monkeypatch):
    """The device codec for every parity / alignment of the data section inside the stream (the section starts wherever the
    names and split k-mers end; rows are n_samples cells, chunks 65 536 bytes): device-written files are read by the host
    codec and the oracle, host-written ones by the device, in groups of 2 chunks."""
    rng = np.random.default_rng(40 + n_samples)
    _, samples = _related_samples(rng, n_samples, length=90_000, snps=200)
    names = [f"s{i}{pad}" for i in range(n_samples)]
    ga = E.DictSet.build([E.record_stream(s) for s in samples], 31, True).merge(names)
    rows = _sorted_export(ga)
    files = {}
    for mode in ("0", "1"):
        set_knob(monkeypatch, "skf_device", mode)
        set_knob(monkeypatch, "skf_group_chunks", "2")
        files[mode] = str(tmp_path / f"w{mode}.skf")
        ga.save(files[mode])
    for written in ("0", "1"):
        for read in ("0", "1"):
            set_knob(monkeypatch, "skf_device", read)
            back = E.Array.load(files[written])
            assert back.names == names
            for x, y in zip(_sorted_export(back), rows):
                assert np.array_equal(x, y), (written, read)
        oa = ora.Array.load(files[written])
        for x, y in zip(_sorted_export(oa), rows):
            assert np.array_equal(x, y), written


def test_write_fasta_streamed_equals_buffer(E, tmp_path):
    """skx_array_write_fasta (pinned double buffers + writer thread) puts out exactly the text skx_array_fasta builds."""
    rng = np.random.default_rng(77)
    _, samples = _related_samples(rng, 9, length=20_000, snps=150)
    names = [f"sample_{i}" * (1 + i % 3) for i in range(9)]
    ga = E.DictSet.build([E.record_stream(s) for s in samples], 31, True).merge(names)
    ga.apply_filters(0.5, False, E.FILTER_NO_CONST, False, False)
    want = ga.fasta()
    p = tmp_path / "aln.fa"
    fd = os.open(str(p), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        ga.write_fasta(fd)
    finally:
        os.close(fd)
    assert p.read_bytes() == want and want.count(b">") == 9


# ---- a snappy frame re-encoder for the decoder test: every element type the format has, chosen at random -------------------
def _crc32c_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def _masked_crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    c ^= 0xFFFFFFFF
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _snappy_decode(block):
    n, shift, ip = 0, 0, 0
    while True:
        b = block[ip]; ip += 1
        n |= (b & 0x7F) << shift; shift += 7
        if not b & 0x80:
            break
    out = bytearray()
    while ip < len(block):
        tag = block[ip]; ip += 1
        t = tag & 3
        if t == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(block[ip:ip + nb], "little") + 1; ip += nb
            out += block[ip:ip + ln]; ip += ln
            continue
        if t == 1:
            ln, off = 4 + ((tag >> 2) & 7), ((tag >> 5) << 8) | block[ip]; ip += 1
        elif t == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(block[ip:ip + 2], "little"); ip += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(block[ip:ip + 4], "little"); ip += 4
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n
    return bytes(out)


def _snappy_encode_random(data, rng, one_literal=False):
    n = len(data)
    out = bytearray()
    v = n
    while True:
        out.append((v & 0x7F) | (0x80 if v >> 7 else 0)); v >>= 7
        if not v:
            break

    def literal(lo, hi):
        ln = hi - lo
        if ln <= 60:
            out.append((ln - 1) << 2)
        else:
            nb = 1 if ln - 1 < 256 else (2 if ln - 1 < 65536 else 3)
            if rng.random() < 0.2 and nb < 4:
                nb += 1                                   # a wider length field than needed is still valid
            out.append((59 + nb) << 2); out.extend((ln - 1).to_bytes(nb, "little"))
        out.extend(data[lo:hi])

    if one_literal:
        literal(0, n)
        return bytes(out)
    offs = [1, 2, 3, 4, 5, 7, 8, 16, 63, 64, 65, 100, 1000, 2046, 2047, 2048, 5000, 40000, 65535]
    i = 0
    while i < n:
        best_len, best_off = 0, 0
        for off in rng.choice(offs, size=6, replace=False):
            off = int(off)
            if off > i:
                continue
            ln = 0
            while ln < 64 and i + ln < n and data[i + ln] == data[i + ln - off]:
                ln += 1
            if ln > best_len:
                best_len, best_off = ln, off
        if best_len >= 1 and rng.random() < 0.9:
            ln = int(rng.integers(1, best_len + 1)) if rng.random() < 0.3 else best_len
            kinds = [2, 4] if best_off < 65536 else [4]
            if 4 <= ln <= 11 and best_off < 2048:
                kinds.append(1)
            kind = int(rng.choice(kinds))
            if kind == 1:
                out.append(1 | ((ln - 4) << 2) | ((best_off >> 8) << 5)); out.append(best_off & 0xFF)
            elif kind == 2:
                out.append(2 | ((ln - 1) << 2)); out.extend(best_off.to_bytes(2, "little"))
            else:
                out.append(3 | ((ln - 1) << 2)); out.extend(best_off.to_bytes(4, "little"))
            i += ln
        else:
            ln = min(n - i, int(rng.choice([1, 2, 3, 17, 60, 61, 64, 200, 300])))
            literal(i, i + ln)
            i += ln
    return bytes(out)


def _reencode_skf(raw, rng):
    """snappy frame -> the same stream, every compressed chunk re-encoded with random (valid) elements."""
    assert raw[:10] == b"\xff\x06\x00\x00sNaPpY"
    out = bytearray(raw[:10])
    i, k = 10, 0
    while i < len(raw):
        typ, ln = raw[i], int.from_bytes(raw[i + 1:i + 4], "little")
        body = raw[i + 4:i + 4 + ln]
        i += 4 + ln
        assert typ in (0, 1)
        data = _snappy_decode(body[4:]) if typ == 0 else bytes(body[4:])
        assert _masked_crc32c(data) == int.from_bytes(body[:4], "little")
        enc = _snappy_encode_random(data, rng, one_literal=(k in (2, 8)))
        assert _snappy_decode(enc) == data
        out += b"\x00" + (len(enc) + 4).to_bytes(3, "little") + _masked_crc32c(data).to_bytes(4, "little") + enc
        k += 1
    return bytes(out)


def test_skf_device_decoder_takes_any_valid_element_stream(E, tmp_path, monkeypatch):
    """Files written by other snappy encoders (the reference's `snap` crate) hold whatever elements their match finder
    chose.  Here every chunk of a file is re-encoded with elements drawn at random from all the format has -- literals with
    0-4 length bytes, copies with 1/2/4-byte offsets, copies overlapping their own output at odd offsets -- and must load on
    the device path to the same array."""
    rng = np.random.default_rng(123)
    _, samples = _related_samples(rng, 5, length=40_000, snps=300)
    names = [f"s{i}" for i in range(5)]
    ga = E.DictSet.build([E.record_stream(s) for s in samples], 31, True).merge(names)
    rows = _sorted_export(ga)
    set_knob(monkeypatch, "skf_device", "0")
    p = str(tmp_path / "host.skf")
    ga.save(p)
    re = _reencode_skf(open(p, "rb").read(), rng)
    p2 = str(tmp_path / "random_elements.skf")
    open(p2, "wb").write(re)
    for mode, group in (("1", "2"), ("1", "8192"), ("0", "8192")):
        set_knob(monkeypatch, "skf_device", mode)
        set_knob(monkeypatch, "skf_group_chunks", group)
        back = E.Array.load(p2)
        assert back.names == names
        for x, y in zip(_sorted_export(back), rows):
            assert np.array_equal(x, y), (mode, group)


@pytest.mark.parametrize("k,length", [(31, 2_000_000), (31, 5_000_000), (31, 6_000_000), (31, 6_600_000), (31, 12_000_000), (31, 25_000_000),
                                      (41, 2_000_000), (41, 5_000_000), (41, 12_000_000)])
def test_dict_every_bucket_configuration(E, k, length):
    """One sample per bucket count / kernel configuration (512, 1 024, 2 048 x 2 dedupe shapes, 4 096, 8 192 buckets: tile size,
    half-tile staging, cursor array and dedupe instantiation all change with it), dictionary bit-exact against the oracle."""
    rng = np.random.default_rng(length % 1000 + 7)
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=length, dtype=np.uint8)]
    g[rng.integers(0, length, size=50)] = ord("N")                       # a few window breaks
    cut = length // 3
    recs = [g[:cut].tobytes(), g[cut:].tobytes()]
    ds = E.DictSet.build([E.record_stream(recs)], k, True)
    ok, ob = oracle_dict(recs, k, True).export()
    gk, gb = ds.export(0)
    assert len(gk) == len(ok)
    assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gk["hi"], ok["hi"]) and np.array_equal(gb, ob)


def test_dedupe_two_shapes_and_listed_regions(E, monkeypatch):
    """Regions are sorted in the launch shape the typical region needs; the few above it are listed and sorted by a second launch of
    the full shape, grid after grid when the list is long.  A 200-base unit repeated 800 times puts ~800 extra words into ~200 of a
    2.5 Mbp sample's 512 regions (mean 4 900 words: 5 x 1 024 for the typical region, 6 x 1 024 for the capacity)."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = acgt[rng.integers(0, 4, size=200)].tobytes()
    samples = []
    for i in range(2):
        body = acgt[rng.integers(0, 4, size=2_340_000)].tobytes()
        samples.append([body[:1_000_000], unit * 800, body[1_000_000:], b"ACGT" * 5])
    set_knob(monkeypatch, "dedupe_spill_grid", "48")               # ~200 listed regions per sample: several second-stage launches
    ds = check_dicts(E, samples, 31, True)
    sizes = [ds.size(i) for i in range(2)]
    ds.free()
    del_knob(monkeypatch, "dedupe_spill_grid")
    ds1 = check_dicts(E, samples, 31, True)                          # one grid of the second stage
    assert [ds1.size(i) for i in range(2)] == sizes
    ds1.free()
