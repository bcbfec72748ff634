"""skx_array_load_filtered (`ska align x.skf` / `ska distance x.skf` in one pass over the file) against load-then-filter through
the general reader and against the oracle (`-m gpu`): same kept rows in the same order, same removed / constant counts, same
alignment bytes, same distance table -- for every filter combination, row widths that are not multiples of anything, groups of
two chunks, and files the one-pass reader must hand to the general path (short keys, stored counts that differ from the rows)."""
import os

import numpy as np
import pytest
from conftest import set_knob, del_knob

import ora

pytestmark = pytest.mark.gpu
ALPHA = np.frombuffer(b"ACGT-ACGTACGT-MRWSYKVHDBN", dtype=np.uint8)


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def _random_array(E, rng, U, S, k=31, small_keys=False):
    lo = rng.integers(1 << 33, 1 << 59, size=U, dtype=np.uint64) if not small_keys else rng.integers(1, 1 << 20, size=U, dtype=np.uint64)
    lo = np.unique(lo)
    rng.shuffle(lo)
    U = len(lo)
    keys = np.zeros(U, E.KEY_DT)
    keys["lo"] = lo
    var = np.empty((U, S), np.uint8)
    kind = rng.integers(0, 5, size=U)
    base = ALPHA[rng.integers(0, 4, size=U)]
    var[:] = base[:, None]                                             # constant rows
    mixed = kind >= 2
    var[mixed] = ALPHA[rng.integers(0, 13, size=(int(mixed.sum()), S))]   # variant rows with gaps
    amb = kind == 4
    var[amb] = ALPHA[rng.integers(0, len(ALPHA), size=(int(amb.sum()), S))]   # ambiguity codes too
    gappy = kind == 1
    g = var[gappy]
    g[rng.random(g.shape) < 0.6] = ord("-")
    var[gappy] = g
    var[rng.integers(0, U, size=max(1, U // 50))] = ord("-")             # rows no sample has
    return keys, var


def _aln_rows(aln):
    return [l for l in aln.split(b"\n")[1::2]]


@pytest.mark.parametrize("S,U", [(1, 70000), (3, 50000), (7, 30000), (64, 9000), (65, 9000), (130, 5000), (257, 3000)])
def test_one_pass_load_equals_load_then_filter(E, tmp_path, monkeypatch, S, U):
    rng = np.random.default_rng(1000 + S)
    keys, var = _random_array(E, rng, U, S)
    names = [f"s{i}" for i in range(S)]
    set_knob(monkeypatch, "skf_device", "1")
    set_knob(monkeypatch, "skf_group_chunks", "2")
    path = str(tmp_path / "a.skf")
    E.Array.from_host(31, True, names, keys, var).save(path)
    oa_full = ora.Array.load(path)
    combos = [(ft, amb, mask, gaps, mf) for ft in range(4) for amb in (False, True) for mask in (False, True) for gaps in (False, True)
              for mf in (0.0, 0.5, 0.9, 1.0)]
    for ft, amb, mask, gaps, mf in combos[:: (1 if S <= 7 else 5)]:
        fast, rem_f, _ = E.Array.load_filtered(path, mf, amb, ft, mask, gaps)
        set_knob(monkeypatch, "no_stream_load", "1")
        slow, rem_s, _ = E.Array.load_filtered(path, mf, amb, ft, mask, gaps)
        del_knob(monkeypatch, "no_stream_load")
        assert (fast.nrows, rem_f) == (slow.nrows, rem_s), (ft, amb, mask, gaps, mf)
        fa, sa = fast.fasta(), slow.fasta()
        assert fa == sa, (ft, amb, mask, gaps, mf)
        oa = ora.Array.load(path)
        assert rem_f == oa.apply_filters(mf, amb, ft, mask, gaps), (ft, amb, mask, gaps, mf)
        # the oracle reads the same file in the same row order, so the alignments agree byte for byte
        assert fa == oa.fasta(), (ft, amb, mask, gaps, mf)
        fast.free(); slow.free()
    assert oa_full.nrows == len(keys)


@pytest.mark.parametrize("S", [2, 6, 40])
@pytest.mark.parametrize("filt_ambig", [True, False])
def test_one_pass_distance_equals_oracle(E, tmp_path, monkeypatch, S, filt_ambig):
    rng = np.random.default_rng(2000 + S)
    keys, var = _random_array(E, rng, 40000 // S + 500, S)
    names = [f"d{i}" for i in range(S)]
    set_knob(monkeypatch, "skf_device", "1")
    set_knob(monkeypatch, "skf_group_chunks", "2")
    path = str(tmp_path / "d.skf")
    E.Array.from_host(31, True, names, keys, var).save(path)
    lib = E.load_library()
    import ctypes as C
    for mf in (0.0, 0.3, 0.9):
        buf, n = C.c_void_p(), C.c_uint64()
        E._check(lib.skh_distance_skf_tsv(E.default_context().h, path.encode(), mf, int(filt_ambig), C.byref(buf), C.byref(n)))
        got = E._take(buf, n)
        assert got == ora.Array.load(path).distance_tsv(mf, filt_ambig), (S, mf, filt_ambig)
        general = E.Array.load(path).distance_tsv(mf, filt_ambig)
        assert got == general


def test_files_the_one_pass_reader_hands_over(E, tmp_path, monkeypatch):
    rng = np.random.default_rng(3)
    set_knob(monkeypatch, "skf_device", "1")
    set_knob(monkeypatch, "skf_group_chunks", "2")
    # (1) keys below 2^32 are shorter CBOR items: the key list is not 9 bytes per key
    keys, var = _random_array(E, rng, 30000, 5, small_keys=True)
    names = [f"s{i}" for i in range(5)]
    p1 = str(tmp_path / "short_keys.skf")
    E.Array.from_host(31, True, names, keys, var).save(p1)
    # (2) stored variant_count differs from what the rows imply (as after `ska weed --filter-ambig-as-missing`)
    keys2, var2 = _random_array(E, rng, 30000, 5)
    counts = (var2 != ord("-")).sum(axis=1).astype(np.uint64)
    counts[::7] = 1
    p2 = str(tmp_path / "counts.skf")
    E.Array.from_host(31, True, names, keys2, var2, counts=counts).save(p2)
    # (3) a k = 41 file (tag-2 bignum keys)
    keys3 = np.zeros(2000, E.KEY_DT)
    keys3["lo"] = rng.integers(1, 1 << 62, size=2000, dtype=np.uint64)
    keys3["hi"] = rng.integers(0, 1 << 16, size=2000, dtype=np.uint64)
    var3 = ALPHA[rng.integers(0, 13, size=(2000, 5))]
    p3 = str(tmp_path / "wide.skf")
    E.Array.from_host(41, True, names, keys3, var3).save(p3)
    for p in (p1, p2, p3):
        for ft, mf in ((1, 0.9), (0, 0.5), (3, 0.0)):
            got, rem, _ = E.Array.load_filtered(p, mf, False, ft, False, False)
            oa = ora.Array.load(p)
            assert rem == oa.apply_filters(mf, False, ft, False, False), (p, ft, mf)
            assert got.fasta() == oa.fasta(), (p, ft, mf)
    # the reference's own fixtures (written by the Rust binary) through the same entry
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for root, _, fs in os.walk(gold):
        for f in fs:
            if f.endswith(".skf"):
                p = os.path.join(root, f)
                got, rem, _ = E.Array.load_filtered(p, 0.9, False, 1, False, False)
                oa = ora.Array.load(p)
                assert rem == oa.apply_filters(0.9, False, 1, False, False)
                assert got.fasta() == oa.fasta(), f


def test_keyless_array_refuses_key_operations(E, tmp_path, monkeypatch):
    rng = np.random.default_rng(4)
    keys, var = _random_array(E, rng, 20000, 4)
    path = str(tmp_path / "k.skf")
    set_knob(monkeypatch, "skf_device", "1")
    E.Array.from_host(31, True, ["a", "b", "c", "d"], keys, var).save(path)
    arr, _, _ = E.Array.load_filtered(path, 0.0, False, 0, False, False)      # nothing removed: rows == split k-mers, but no keys
    with pytest.raises(E.EngineError):
        arr.save(str(tmp_path / "again.skf"))
    with pytest.raises(E.EngineError):
        arr.export()


def _reframe_small_chunks(raw, rng, lo, hi):
    """a snappy frame -> the same stream cut into chunks of lo..hi uncompressed bytes (what another framer, or a writer that flushes
    often, produces): compressed and uncompressed chunks mixed, each with its own masked CRC-32C"""
    from test_gpu_parity import _masked_crc32c, _snappy_decode, _snappy_encode_random
    assert raw[:10] == b"\xff\x06\x00\x00sNaPpY"
    data = bytearray()
    i = 10
    while i < len(raw):
        typ, ln = raw[i], int.from_bytes(raw[i + 1:i + 4], "little")
        body = raw[i + 4:i + 4 + ln]
        i += 4 + ln
        data += _snappy_decode(body[4:]) if typ == 0 else bytes(body[4:])
    out = bytearray(raw[:10])
    p = 0
    while p < len(data):
        n = int(rng.integers(lo, hi + 1))
        piece = bytes(data[p:p + n])
        p += n
        if rng.random() < 0.5:
            enc = _snappy_encode_random(piece, rng)
            out += b"\x00" + (len(enc) + 4).to_bytes(3, "little") + _masked_crc32c(piece).to_bytes(4, "little") + enc
        else:
            out += b"\x01" + (len(piece) + 4).to_bytes(3, "little") + _masked_crc32c(piece).to_bytes(4, "little") + piece
    return bytes(out)


@pytest.mark.parametrize("lo,hi", [(200, 3000), (1, 40), (30000, 65536)])
def test_files_framed_in_small_chunks(E, tmp_path, monkeypatch, lo, hi):
    """A valid .skf whose snappy chunks are smaller than 64 KB has more chunks than its size suggests: the one-pass reader takes them
    in more groups (its buffers are sized for 64 KB chunks) or hands the file to the general reader when its chunk directory runs
    out; either way the array, the alignment and the distances are those of the original file and of the oracle."""
    rng = np.random.default_rng(lo * 7 + hi)
    S, U = 5, (8000 if hi <= 40 else 60000)            # tiny chunks: more of them than the asynchronous walk reserves
    keys, var = _random_array(E, rng, U, S)
    names = [f"s{i}" for i in range(S)]
    path, small = str(tmp_path / "a.skf"), str(tmp_path / "small.skf")
    E.Array.from_host(31, True, names, keys, var).save(path)
    open(small, "wb").write(_reframe_small_chunks(open(path, "rb").read(), rng, lo, hi))
    assert ora.Array.load(small).nrows == len(keys)                    # a valid file: the oracle's reader takes it
    for group in ("8192", "3"):
        set_knob(monkeypatch, "skf_group_chunks", group)
        for ft, mf in ((E.FILTER_NO_CONST, 0.9), (E.FILTER_NONE, 0.0), (E.FILTER_NO_AMBIG_OR_CONST, 0.5)):
            a, rem_a, _ = E.Array.load_filtered(path, mf, False, ft, False, False)
            b, rem_b, _ = E.Array.load_filtered(small, mf, False, ft, False, False)
            assert rem_a == rem_b and a.fasta() == b.fasta(), (group, ft, mf)
            oa = ora.Array.load(small)
            assert rem_b == oa.apply_filters(mf, False, ft, False, False) and b.fasta() == oa.fasta()
            a.free(); b.free()
        whole = E.Array.load(small)                                     # the general loader
        k1, v1, c1 = whole.export()
        k0, v0, c0 = E.Array.load(path).export()
        assert np.array_equal(k1, k0) and np.array_equal(v1, v0) and np.array_equal(c1, c0)
        d1 = E.Array.load(small).distance_tsv(0.0, True)
        assert d1 == ora.Array.load(small).distance_tsv(0.0, True)


@pytest.mark.parametrize("S,U", [(4, 70000), (33, 9000)])
def test_one_pass_load_with_128_bit_keys(E, tmp_path, monkeypatch, S, U):
    """k = 41 (lib.rs:635-661: the u128 file): the split k-mer list holds items of every length -- small uints, 9-byte uints, tag-2
    bignums of 9 to 11 bytes -- and is walked, not jumped over; the rows then stream through the same one-pass load + filter as 64-bit
    files: == load-then-filter == the oracle, for `align`'s filters and `distance`'s two stages."""
    rng = np.random.default_rng(4100 + S)
    keys, var = _random_array(E, rng, U, S)
    n = len(keys)
    kind = rng.integers(0, 6, size=n)
    keys["hi"] = np.where(kind >= 2, rng.integers(1, 1 << 16, size=n, dtype=np.uint64), 0)      # most beyond 64 bits
    keys["lo"][kind == 0] = np.unique(rng.integers(0, 1 << 30, size=n * 2, dtype=np.uint64))[: int((kind == 0).sum())] | np.uint64(1 << 31)   # short uints
    keys["lo"][:4] = [3, 200, 60000, 1 << 40]                          # 1-, 2-, 3- and 9-byte items
    keys["hi"][:4] = 0
    _, first = np.unique(keys, return_index=True)                      # distinct (hi, lo) pairs only
    keys, var = keys[np.sort(first)], var[np.sort(first)]
    names = [f"s{i}" for i in range(S)]
    set_knob(monkeypatch, "skf_device", "1")
    set_knob(monkeypatch, "skf_group_chunks", "2")
    path = str(tmp_path / "wide.skf")
    E.Array.from_host(41, True, names, keys, var).save(path)
    assert ora.Array.load(path).nrows == len(keys)
    E.phases(reset=True)
    for ft, amb, mask, gaps, mf in [(1, False, False, False, 0.9), (0, False, False, False, 0.0), (3, True, True, False, 0.5), (2, False, True, True, 1.0)]:
        fast, rem_f, _ = E.Array.load_filtered(path, mf, amb, ft, mask, gaps)
        set_knob(monkeypatch, "no_stream_load", "1")
        slow, rem_s, _ = E.Array.load_filtered(path, mf, amb, ft, mask, gaps)
        del_knob(monkeypatch, "no_stream_load")
        oa = ora.Array.load(path)
        assert rem_f == rem_s == oa.apply_filters(mf, amb, ft, mask, gaps), (ft, amb, mask, gaps, mf)
        assert fast.fasta() == slow.fasta() == oa.fasta(), (ft, amb, mask, gaps, mf)
        fast.free(); slow.free()
    assert "load.stream_decode_filter" in E.phases()                   # the one-pass reader took the file
    for mf, filt in ((0.0, True), (0.6, False)):
        a, _, constant = E.Array.load_filtered(path, mf, False, 1, False, False, two_stage=True)
        d = a.distance(constant, filt)
        oa = ora.Array.load(path)
        if mf * S >= 1.0:
            oa.filter(int(np.ceil(S * mf)), False, ora.FILTER_NONE, False, False, False)
        oc = oa.filter(0, False, ora.FILTER_NO_CONST, False, False, False)
        od = oa.distance(oc, filt)
        assert constant == oc and np.array_equal(d["match_count"], od["match_count"]) and np.array_equal(d["mismatch_count"], od["mismatch_count"])
        assert np.allclose(d["distance"], od["distance"], rtol=0, atol=1e-6)
