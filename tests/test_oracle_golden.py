"""Pins the CPU oracle against the reference's own golden vectors."""
import numpy as np
import pytest

import golden_cases as G
import ora


@pytest.mark.parametrize("case", G.ALL_CASES, ids=lambda c: c.__name__)
def test_reference_case(case, tmp_path):
    case(ora, tmp_path)


def _as_map(arr):
    keys, var, counts = arr.export()
    return {(int(k["hi"]) << 64) | int(k["lo"]): (bytes(v), int(c)) for k, v, c in zip(keys, var, counts)}


@pytest.mark.parametrize("fixture,k,files", [
    ("merge.skf", 17, ["test_1.fa", "test_2.fa"]),
    ("merge_k9.skf", 9, ["test_1.fa", "test_2.fa"]),
    ("merge_k41.skf", 41, ["test_1.fa", "test_2.fa"]),
    ("multidist.skf", 9, ["N_test_1.fa", "N_test_2.fa", "ambig_test_1.fa", "ambig_test_2.fa", "test_1.fa", "test_2.fa"]),
])
def test_skf_fixture_reproduced(fixture, k, files):
    """The four .skf files written by the Rust binary == our build, as exact {k-mer -> (row, count)} maps."""
    ref = ora.Array.load(G.fin(fixture))
    mine = ora.Array.build(G.fasta_inputs(ora, [G.fin(f) for f in files]), k=k)
    assert ref.k == mine.k == k and ref.rc == mine.rc and ref.k_bits == mine.k_bits
    assert ref.names == mine.names
    assert _as_map(ref) == _as_map(mine)
    assert ref.nkmers == {"merge.skf": 78, "merge_k9.skf": 68, "merge_k41.skf": 43, "multidist.skf": 89}[fixture]


def test_skf_roundtrip(tmp_path):
    for f in ("merge.skf", "merge_k41.skf", "multidist.skf"):
        a = ora.Array.load(G.fin(f))
        p = str(tmp_path / f)
        a.save(p)
        b = ora.Array.load(p)
        assert _as_map(a) == _as_map(b) and a.names == b.names and a.version == b.version


@pytest.mark.parametrize("fixture", ["merge.skf", "merge_k9.skf", "merge_k41.skf", "multidist.skf"])
def test_skf_writer_reproduces_the_rust_written_bytes(fixture, tmp_path):
    """merge_ska_array.rs:108-126,191-197: load -> save of a file the Rust binary wrote gives back its CBOR document byte for byte (field
    order, definite lengths, minimal-length integers, tag-2 bignums for keys above 64 bits, 2 bytes per cell above 23).  The snappy framing
    around it is free (any valid frame loads), so the comparison is of the un-framed streams."""
    a = ora.Array.load(G.fin(fixture))
    p = str(tmp_path / fixture)
    a.save(p)
    want, got = ora.skf_cbor(G.fin(fixture)), ora.skf_cbor(p)
    assert len(want) == {"merge.skf": 894, "merge_k9.skf": 655, "merge_k41.skf": 843, "multidist.skf": 1577}[fixture]
    assert got == want


def test_load_u64_then_u128():
    # lib.rs:635-661: a k=41 file must fail as u64 and load as u128
    with pytest.raises(ora.OracleError):
        ora.Array.load(G.fin("merge_k41.skf"), want_bits=64)
    assert ora.Array.load(G.fin("merge_k41.skf"), want_bits=128).k_bits == 128


def test_sample_names():
    assert ora.sample_name("/a/b/test_1.fa") == "test_1"
    assert ora.sample_name("x.fastq.gz") == "x"
    assert ora.sample_name("dir/x.FASTA") == "x"
    assert ora.sample_name("dir/x.fa.gz") == "dir/x.fa.gz"
    assert ora.sample_name("x.fasta.fa") == "x.fasta"


def test_end_quirk_q1():
    """split_kmer.rs:89,121: a (re)start needs idx + k < len (strict)."""
    k = 7
    assert len(ora.extract_record(b"ACGTACG", k)[0]) == 0            # L == k -> nothing
    assert len(ora.extract_record(b"ACGTACGT", k)[0]) == 2           # L == k+1 -> two windows
    assert len(ora.extract_record(b"ACGTACGTNACGTACG", k)[0]) == 2   # trailing clean run of exactly k dropped
    assert len(ora.extract_record(b"ACGTACGTNACGTACGA", k)[0]) == 4  # k+1 trailing run kept
    assert len(ora.extract_record(b"ACGTACGNACGTACGAA", k)[0]) == 4  # run of exactly k before a bad base kept


def test_nthash_roll_equals_recompute():
    rng = np.random.default_rng(0)
    seq = bytes(rng.choice(list(b"ACGT"), size=200).tolist())
    for k in (7, 31, 33, 63):
        for rc in (True, False):
            _, _, _, h = ora.extract_record(seq, k, rc=rc, qual_bytes=b"I" * len(seq), is_reads=True)
            for i in (0, 1, 50, len(h) - 2):
                _, _, _, h1 = ora.extract_record(seq[i:i + k + 1], k, rc=rc, qual_bytes=b"I" * (k + 1), is_reads=True)
                assert h[i] == h1[0]


def test_distance_modes_on_rows_without_ambiguous_cells(tmp_path):
    """What the engine's --allow-ambiguous row split rests on, stated on the oracle alone (merge_ska_array.rs:587-632 sums per row, so rows
    can be counted in any grouping): on rows without an ambiguous cell the three counts of the default sweep -- exactly one cell missing,
    both present (= both unambiguous there), both present and equal -- are all the twelve-class form needs.  In the reference's own
    outputs: same distance and mismatch count in both modes, and the allow-ambiguous "matches" (cells that share a base) are the default
    mode's "matches" (cells compared) minus the differing ones."""
    import numpy as np
    rng = np.random.default_rng(11)
    anc = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=6000)]
    inputs = []
    for i in range(7):
        s = anc.copy()
        pos = rng.integers(0, len(s), size=40)
        s[pos] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=40)]
        if i % 3 == 2:
            s = s[:4000]                                        # missing k-mers
        p = tmp_path / f"c{i}.fa"
        p.write_bytes(b">c\n" + s.tobytes() + b"\n")
        inputs.append((f"c{i}", str(p), None))
    a = ora.Array.build(inputs, k=21)
    _, v, _ = a.export()
    assert set(np.unique(v).tolist()) <= set(b"-ACGT") and (v == ord("-")).any()      # no ambiguity codes in this set
    c = a.filter(0, False, ora.FILTER_NO_CONST, False, False, False)
    f, g = a.distance(c, True), a.distance(c, False)
    assert np.array_equal(f["mismatch_count"], g["mismatch_count"])
    assert np.allclose(f["distance"], g["distance"], rtol=0, atol=1e-9) and (f["distance"] > 0).any()
    assert np.array_equal(g["match_count"], f["match_count"] - np.rint(f["distance"]).astype(np.uint64))
