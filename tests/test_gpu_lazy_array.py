"""Lazily held arrays (`-m gpu`): skx_build_and_merge returns rows + dictionaries, not a matrix.  `ska build` streams such an
array into its .skf window by window, `ska align *.fa` filters it before any cell is written; every other operation assembles
the matrix first.  All of them must give what the eagerly assembled array (SKX_KNOBS=eager_array) and the oracle give."""
import os

import numpy as np
import pytest
from conftest import set_knob, del_knob

import ora

pytestmark = pytest.mark.gpu
FILTERS = [(ft, amb, mask, gaps) for ft in range(4) for amb in (False, True) for mask in (False, True) for gaps in (False, True)]


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def _files(tmp_path, n=9, length=60_000, snps=120, seed=3):
    rng = np.random.default_rng(seed)
    anc = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=length)]
    inputs = []
    for i in range(n):
        s = anc.copy()
        pos = rng.integers(0, length, size=snps)
        s[pos] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=snps)]
        if i % 3 == 2:
            s = s[: length * 2 // 3]                            # missing k-mers
        seq = s.tobytes()
        if i % 4 == 1:
            seq = seq + b"\n>rep\n" + seq[1000:1400]            # a repeat with changed flanks -> ambiguity codes
            seq = seq[:1200] + b"T" + seq[1201:]
        p = tmp_path / f"s{i}.fa"
        with open(p, "wb") as f:
            f.write(b">c0\n" + seq[: len(seq) // 2] + b"\n>c1\n" + seq[len(seq) // 2:] + b"\n")
        inputs.append((f"s{i}", str(p), None))
    return inputs


def _sorted(arr):
    k, v, c = arr.export()
    o = np.lexsort((k["lo"], k["hi"]))                          # 128-bit keys (k > 31) order by (hi, lo)
    return k["lo"][o], k["hi"][o], v[o], c[o]


@pytest.mark.parametrize("k", [31, 41])                        # 64- and 128-bit keys: the same lazily held form (lib.rs:592-622)
def test_lazy_save_streams_the_same_file_rows(E, tmp_path, monkeypatch, k):
    inputs = _files(tmp_path)
    oa = ora.Array.build(inputs, k=k)
    want = _sorted(oa)
    set_knob(monkeypatch, "skf_device", "1")
    set_knob(monkeypatch, "skf_group_chunks", "2")              # several windows
    lazy = E.Array.build(inputs, k=k, threads=3)
    assert list(lazy.sample_kmers()) == [int(x) for x in (oa.export()[1] != ord("-")).sum(axis=0)]      # answered from the dictionaries
    p = str(tmp_path / "lazy.skf")
    lazy.save(p)                                                 # still lazy: windows
    back = ora.Array.load(p)
    for x, y in zip(_sorted(back), want):
        assert np.array_equal(x, y)
    set_knob(monkeypatch, "skf_device", "0")                    # host codec: rows fetched block by block
    p2 = str(tmp_path / "lazy_host.skf")
    E.Array.build(inputs, k=k, threads=3).save(p2)
    for x, y in zip(_sorted(ora.Array.load(p2)), want):
        assert np.array_equal(x, y)
    # and the materialised array is the same array
    for x, y in zip(_sorted(lazy), want):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("k", [15, 41])
@pytest.mark.parametrize("min_freq", [0.0, 0.5, 0.9, 1.0])
def test_lazy_filter_writes_only_kept_rows(E, tmp_path, monkeypatch, min_freq, k):
    inputs = _files(tmp_path, n=7, length=30_000, seed=11)
    for ft, amb, mask, gaps in (FILTERS if k == 15 else FILTERS[::3]):
        lazy = E.Array.build(inputs, k=k, threads=2)
        g = lazy.align(filter_type=ft, mask_ambig=mask, ignore_const_gaps=gaps, min_freq=min_freq, filter_ambig_as_missing=amb)
        set_knob(monkeypatch, "eager_array", "1")
        eager = E.Array.build(inputs, k=k, threads=2)
        del_knob(monkeypatch, "eager_array")
        e = eager.align(filter_type=ft, mask_ambig=mask, ignore_const_gaps=gaps, min_freq=min_freq, filter_ambig_as_missing=amb)
        assert g == e, (ft, amb, mask, gaps)                     # same engine order either way: byte-identical
        oa = ora.Array.build(inputs, k=k)
        o = oa.align(filter_type=ft, mask_ambig=mask, ignore_const_gaps=gaps, min_freq=min_freq, filter_ambig_as_missing=amb)
        assert sorted(zip(*g.decode().splitlines()[1::2])) == sorted(zip(*o.decode().splitlines()[1::2])), (ft, amb, mask, gaps)


@pytest.mark.parametrize("k", [21, 41])
def test_lazy_filter_then_export_and_counts(E, tmp_path, k):
    inputs = _files(tmp_path, n=6, length=25_000, seed=21)
    for ft, amb, mask, gaps in FILTERS[::5]:
        lazy = E.Array.build(inputs, k=k, threads=2)
        oa = ora.Array.build(inputs, k=k)
        assert lazy.filter(3, amb, ft, mask, gaps, True) == oa.filter(3, amb, ft, mask, gaps, True)
        for x, y in zip(_sorted(lazy), _sorted(oa)):
            assert np.array_equal(x, y), (ft, amb, mask, gaps)


def test_lazy_array_other_operations_materialise(E, tmp_path):
    inputs = _files(tmp_path, n=5, length=20_000, seed=31)
    oa = ora.Array.build(inputs, k=31)
    assert E.Array.build(inputs, k=31).distance_tsv() == ora.Array.build(inputs, k=31).distance_tsv()
    assert E.Array.build(inputs, k=31).nk(full_info=True) == oa.nk(full_info=True)
    a = E.Array.build(inputs, k=31)
    a.delete_samples(["s1", "s3"])
    o = ora.Array.build(inputs, k=31)
    o.delete_samples(["s1", "s3"])
    for x, y in zip(_sorted(a), _sorted(o)):
        assert np.array_equal(x, y)
    m = E.Array.merge([E.Array.build(inputs[:2], k=31), E.Array.build(inputs[2:], k=31)])
    for x, y in zip(_sorted(m), _sorted(oa)):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("k", [31, 41])
def test_assemble_lazy_equals_assemble(E, tmp_path, k):
    """skx_array_assemble_lazy (the multi-GPU ranks' form: rows from elsewhere + the local dictionaries, no matrix) == skx_array_assemble:
    statistics without a matrix, the filter writing kept rows only, the streamed .skf, the full export."""
    inputs = _files(tmp_path, n=6, length=40_000, seed=41)
    streams = [open(p, "rb").read() for _, p, _ in inputs]
    recs = [b"\n".join(l for l in s.split(b"\n") if l and not l.startswith(b">")) + b"\n" for s in streams]
    names = [nm for nm, _, _ in inputs]

    def both():
        ds = E.DictSet.build(recs, k, True)
        rows = ds.union_keys()
        return ds, rows
    ds, rows = both()
    eager = ds.assemble(rows, names)
    ds2, rows2 = both()
    lazy = ds2.assemble_lazy(rows2, names)
    assert ds2.h is None and rows2.h is None                        # both passed into the array
    import torch  # noqa: F401  (device_stats hands out device pointers; compare through the exports instead)
    for x, y in zip(_sorted(lazy), _sorted(eager)):                 # export materialises the lazy one
        assert np.array_equal(x, y)
    for ft, amb, mask, gaps in FILTERS[::7]:
        ds3, rows3 = both()
        lz = ds3.assemble_lazy(rows3, names)
        lz.device_stats()                                           # statistics only: still no matrix
        ds4, rows4 = both()
        eg = ds4.assemble(rows4, names)
        assert lz.filter(3, amb, ft, mask, gaps, True) == eg.filter(3, amb, ft, mask, gaps, True)
        for x, y in zip(_sorted(lz), _sorted(eg)):
            assert np.array_equal(x, y), (ft, amb, mask, gaps)
    ds5, rows5 = both()
    p = str(tmp_path / "lazy_assembled.skf")
    ds5.assemble_lazy(rows5, names).save(p)
    for x, y in zip(_sorted(ora.Array.load(p)), _sorted(eager)):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("filt", [True, False])
@pytest.mark.parametrize("min_freq", [0.0, 0.6])
def test_distance_filtered_leaves_the_array_alone(E, tmp_path, filt, min_freq):
    """skx_array_distance_filtered: generic_modes::distance's two filters decide per row, the bit planes are built over the rows that
    stay -- same pairs as filter + filter + distance on the oracle, and the array afterwards is the array before."""
    import math
    inputs = _files(tmp_path, n=9, length=50_000, seed=51)
    arr = E.Array.build(inputs, k=31, threads=2)
    before = _sorted(arr)
    got, constant, rows_used = arr.distance_filtered(min_freq, filt)
    oa = ora.Array.build(inputs, k=31)
    if min_freq * 9 >= 1.0:
        oa.filter(math.ceil(9 * min_freq), False, ora.FILTER_NONE, False, False, False)
    oc = oa.filter(0, False, ora.FILTER_NO_CONST, False, False, False)
    od = oa.distance(oc, filt)
    assert constant == oc and rows_used == oa.nrows
    assert np.array_equal(got["match_count"], od["match_count"]) and np.array_equal(got["mismatch_count"], od["mismatch_count"])
    assert np.allclose(got["distance"], od["distance"], rtol=0, atol=1e-6)
    for x, y in zip(_sorted(arr), before):
        assert np.array_equal(x, y)


def test_phase_recorder_lists_what_ran(E, tmp_path):
    inputs = _files(tmp_path, n=3, length=20_000, seed=61)
    E.phases(reset=True)
    a = E.Array.build(inputs, k=31, threads=2)
    a.save(str(tmp_path / "p.skf"))
    ph = E.phases()
    assert {"build.read_upload", "build.dictionaries", "save.data_section"} <= set(ph) and all(v >= 0 for v in ph.values())
    assert E.phases(reset=True) and not E.phases()


@pytest.mark.parametrize("filt", [True, False])
def test_distance_many_samples_several_pair_tiles(E, tmp_path, filt):
    """131 samples: three tile rows / columns of the 64 x 64 (default) and five of the 32 x 32 (--allow-ambiguous) pair sweep, the last
    ones ragged, ambiguity codes present -- every pair as the oracle counts it."""
    inputs = _files(tmp_path, n=131, length=6_000, snps=40, seed=71)
    arr = E.Array.build(inputs, k=21, threads=4)
    got, constant, _ = arr.distance_filtered(0.0, filt)
    oa = ora.Array.build(inputs, k=21)
    oc = oa.filter(0, False, ora.FILTER_NO_CONST, False, False, False)
    od = oa.distance(oc, filt)
    assert constant == oc and len(got) == 131 * 130 // 2
    assert np.array_equal(got["match_count"], od["match_count"]) and np.array_equal(got["mismatch_count"], od["mismatch_count"])
    assert np.allclose(got["distance"], od["distance"], rtol=0, atol=1e-6)
    assert np.allclose(got["mismatch_prop"], od["mismatch_prop"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("filt", [True, False])
def test_bit_planes_are_the_cells(E, tmp_path, filt):
    """The bit planes the pair sweep reads (skx_array_distance_planes: 4 planes when ambiguous cells are filtered, 8 with
    --allow-ambiguous), bit for bit against the cells of the array: row r of sample s is bit r % 64 of word r / 64; the padding of the
    last word is zero.  Ambiguity codes and missing cells are present; 11 samples = one full group of eight per workgroup and a ragged one."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")                       # the runtime the engine itself is linked against (already mapped)
    inputs = _files(tmp_path, n=11, length=50_000, snps=150, seed=29)
    arr = E.Array.build(inputs, k=31, threads=4)
    mp, pitch, U = arr.device_matrix()                          # the cells as the plane kernel reads them: [samples][pitch], row r of sample s at s * pitch + r
    S = arr.nsamples
    cells = np.zeros((S, pitch), np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(cells.ctypes.data), ctypes.c_void_p(int(mp)), ctypes.c_size_t(cells.nbytes), 2) == 0     # hipMemcpyDeviceToHost
    v = cells[:, :U].T
    assert U > 3 * 4096 and len(set(np.unique(v).tolist()) - set(b"-ACGT")) > 0
    p, wpr, n_planes = arr.distance_planes(filt)
    assert wpr == (U + 63) // 64 and n_planes == (4 if filt else 8)
    got = np.zeros((n_planes, S, wpr), np.uint64)
    assert hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(int(p)), ctypes.c_size_t(got.nbytes), 2) == 0      # hipMemcpyDeviceToHost
    code = np.full(256, 15, np.uint8)                           # set codes of the IUPAC letters: A 1, C 2, T 4, G 8 and their unions
    for ch, c in zip(b"-ACMTWYHGRSVKDBN", range(16)):
        code[ch] = c
    c = code[v.T]                                               # [samples][rows]
    size = np.array([bin(x).count("1") for x in range(16)], np.uint8)[c]
    if filt:
        bits = [c != 0, size == 1, (size == 1) & ((c & 10) != 0), (size == 1) & ((c & 12) != 0)]
    else:
        bits = [c != 0, (c & 1) != 0, (c & 2) != 0, (c & 4) != 0, (c & 8) != 0, size == 1, size == 2, size == 3]
    for pl, b in enumerate(bits):
        padded = np.zeros((S, wpr * 64), np.uint8)
        padded[:, :U] = b
        want = np.packbits(padded.reshape(S, wpr, 64), axis=2, bitorder="little").view(np.uint64).reshape(S, wpr)
        assert np.array_equal(got[pl], want), pl


def test_allow_ambiguous_does_not_trust_stale_row_statistics(E, tmp_path, monkeypatch):
    """--allow-ambiguous sends the rows whose statistics show no ambiguous cell through the three-count sweep.  With statistics that missed
    the ambiguity codes (forced here: every kept row passed off as clean) the planes give it away -- present != unambiguous somewhere --
    and every row takes the twelve-class sweep: same table as the oracle's either way."""
    inputs = _files(tmp_path, n=13, length=20_000, snps=60, seed=5)
    oa = ora.Array.build(inputs, k=21)
    oc = oa.filter(0, False, ora.FILTER_NO_CONST, False, False, False)
    od = oa.distance(oc, False)
    for stale in (False, True):
        if stale:
            set_knob(monkeypatch, "stale_row_mask", "1")
        arr = E.Array.build(inputs, k=21, threads=4)
        got, constant, _ = arr.distance_filtered(0.0, False)
        assert constant == oc
        assert np.array_equal(got["match_count"], od["match_count"]) and np.array_equal(got["mismatch_count"], od["mismatch_count"])
        assert np.allclose(got["distance"], od["distance"], rtol=0, atol=1e-6)
