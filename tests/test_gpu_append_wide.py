"""MergeSkaDict<u128>::append on the device (skx_append_wide.inc; reference: merge_ska_dict.rs:28-39,77-109, lib.rs:592-622 for the
choice of u128 above k = 31): the one-pass merge of 128-bit assemblies against the CPU oracle, with the path taken asserted
(`skx_ctx_merge_path`).  Bit-exact: split k-mers, middle bases, counts, alignments."""
import numpy as np
import pytest
from conftest import set_knob

import ora
from test_gpu_parity import as_map, build_both, rand_records

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def related(rng, length, n, snps, cut=2):
    anc = ACGT[rng.integers(0, 4, size=length)]
    out = []
    for _ in range(n):
        s = anc.copy()
        pos = rng.integers(0, length, size=snps)
        s[pos] = ACGT[rng.integers(0, 4, size=snps)]
        b = s.tobytes()
        step = (length + cut - 1) // cut
        out.append([b[i:i + step] for i in range(0, length, step)])
    return out


def check_equal(E, ga, oa):
    assert ga.names == oa.names and ga.nkmers == oa.nkmers
    assert as_map(*ga.export()) == as_map(*oa.export())
    assert list(ga.sample_kmers()) == [int(x) for x in (oa.export()[1] != ord("-")).sum(axis=0)]


# (k, genome length): the kernel is instantiated per (dword pair of the key shift, dword of the part bits); these sizes give every pair
# that the host can choose -- rem = 2 (k - 1) - logQ, logQ from the number of rows:
#   (33, 3 000) rem 64: shift 59, part bits in dword 2      (33, 70 000) rem 59: shift 64, part bits start at bit 63
#   (33, 150 000) rem 58: shift 65, part bits in dword 1     (47, 5 000) rem 91: shift 32, part bits start at bit 95
#   (63, 4 000) rem 113: shift 10, part bits in dword 3      (41, 20 000) rem 77     (55, 9 000) rem 106
@pytest.mark.parametrize("k,length", [(33, 3000), (33, 70_000), (33, 150_000), (47, 5000), (63, 4000), (41, 20_000), (55, 9000), (35, 40_000)])
@pytest.mark.parametrize("rc", [True, False])
def test_wide_append_every_shift_shape(E, k, length, rc):
    rng = np.random.default_rng(1000 * k + length)
    samples = related(rng, length, 5, 20)
    samples.append(samples[0][:1] + rand_records(rng, 2, 400))        # a sample that holds part of the rows only
    ga, oa = build_both(E, samples, k, rc)
    assert E.default_context().merge_path() == "append128"
    check_equal(E, ga, oa)


@pytest.mark.parametrize("k", [33, 41, 63])
def test_wide_append_many_samples_repeats_and_palindromes(E, k):
    """More samples than the sixteen waves of a row block (each wave takes several), samples that are one repeat (every word of a load
    in one row block: the 128-bit pass's queue takes it, unlike the 64-bit one's), ambiguity where copies differ, palindromic middles."""
    rng = np.random.default_rng(7 + k)
    samples = related(rng, 30_000, 37, 15, cut=3)
    unit = ACGT[rng.integers(0, 4, size=k + 9)].tobytes()
    # (small enough for the regions' fixed capacity: a repeat that overflows a region sends the build to exact offsets and sorted dictionaries)
    samples += [[b"A" * 700, unit * 8], [b"AT" * 300 + samples[0][0][:3000]], [b"ACGT" * 150]]
    var = bytearray(unit * 50)
    for p in range(k // 2, len(var), len(unit)):
        var[p] = b"ACGT"[(p // len(unit)) % 4]                          # the same flanks around different middle bases: folded into ambiguity codes
    samples.append([bytes(var)])
    ga, oa = build_both(E, samples, k, True)
    assert E.default_context().merge_path() == "append128"
    check_equal(E, ga, oa)
    # the kept rows straight from the pieces (pieces_rows_kernel<kept>), the filters' verdicts from pieces_stats_kernel's counts, distances
    for min_freq, ft, amb in ((0.0, 0, False), (0.5, 1, True), (0.9, 2, False), (1.0, 3, True)):
        ga, oa = build_both(E, samples, k, True)
        g = ga.align(filter_type=ft, min_freq=min_freq, filter_ambig_as_missing=amb, mask_ambig=amb)
        o = oa.align(filter_type=ft, min_freq=min_freq, filter_ambig_as_missing=amb, mask_ambig=amb)
        assert sorted(zip(*g.decode().splitlines()[1::2])) == sorted(zip(*o.decode().splitlines()[1::2])), (min_freq, ft, amb)
    ga, oa = build_both(E, samples[:12], k, True)
    assert ga.distance_tsv(min_freq=0.5) == oa.distance_tsv(min_freq=0.5)


@pytest.mark.parametrize("k", [33, 63])
def test_wide_repeat_that_overflows_a_region_takes_the_sorted_path(E, k):
    rng = np.random.default_rng(70 + k)
    samples = related(rng, 30_000, 6, 15) + [[b"A" * 20_000, b"AC" * 9000]]
    ga, oa = build_both(E, samples, k, True)
    assert E.default_context().merge_path().startswith("sorted: the dictionaries are sorted")
    check_equal(E, ga, oa)


@pytest.mark.parametrize("k", [41])
def test_wide_append_equals_the_sorted_path(E, k, monkeypatch):
    """The same samples through the pass and through per-sample sort + union + assemble (SKX_KNOBS=sorted_wide): one array."""
    rng = np.random.default_rng(3)
    samples = related(rng, 50_000, 9, 30)
    ga, oa = build_both(E, samples, k, True)
    assert E.default_context().merge_path() == "append128"
    m1 = as_map(*ga.export())
    set_knob(monkeypatch, "sorted_wide", 1)
    gb, _ = build_both(E, samples, k, True)
    assert E.default_context().merge_path().startswith("sorted:")
    assert as_map(*gb.export()) == m1 == as_map(*oa.export())


def test_merge_path_is_reported_for_64_bit_keys(E):
    rng = np.random.default_rng(11)
    samples = related(rng, 20_000, 20, 10)
    ga, oa = build_both(E, samples, 31, True)
    assert E.default_context().merge_path() == "append64"
    check_equal(E, ga, oa)


# ---- the .skf writer, byte for byte (merge_ska_array.rs:108-126,191-204: ciborium over the struct, snap's frame around it) ----
import golden_cases as G


@pytest.mark.parametrize("fixture", ["merge.skf", "merge_k9.skf", "merge_k41.skf", "multidist.skf"])
def test_engine_rewrites_a_rust_written_skf_byte_for_byte(E, fixture, tmp_path):
    """A file the Rust binary wrote, loaded through skx_array_load and saved again: rows stay in file order and ska_version is the file's, so
    the CBOR document must come out identical -- field order, definite lengths, minimal-length integers, tag-2 bignums for 128-bit keys,
    two bytes per cell.  (The snappy frame around it is free: any valid frame loads; the engine compresses, the fixture's writer did too.)"""
    a = E.Array.load(G.fin(fixture))
    p = str(tmp_path / fixture)
    a.save(p)
    assert ora.skf_cbor(p) == ora.skf_cbor(G.fin(fixture))
    # and the streaming writer of `ska build` (skh_save_skf: device-encoded data section)
    b = E.Array.load(G.fin(fixture))
    b.save_skf(str(tmp_path / ("s_" + fixture[:-4])))
    assert ora.skf_cbor(str(tmp_path / ("s_" + fixture))) == ora.skf_cbor(G.fin(fixture))


@pytest.mark.parametrize("k", [31, 41, 63])
def test_engine_built_skf_equals_the_oracle_writer_on_the_same_rows(E, k, tmp_path):
    """An array built here, written by the engine; the oracle reads it (rows in file order) and writes it with its own writer -- the one
    test_oracle_golden.py pins against the Rust-written fixtures byte for byte: the two CBOR documents must be identical."""
    rng = np.random.default_rng(k)
    samples = related(rng, 40_000, 11, 25)
    names = [f"s{i}" for i in range(len(samples))]
    ga = E.DictSet.build([E.record_stream(r) for r in samples], k, True).merge(names)
    p, q = str(tmp_path / "e.skf"), str(tmp_path / "o.skf")
    ga.save(p)
    o = ora.Array.load(p)
    o.save(q)
    assert ora.skf_cbor(p) == ora.skf_cbor(q)
    assert o.version == ga.version if hasattr(ga, "version") else True
