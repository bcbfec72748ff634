"""FASTA text -> record stream on the device (skx_parse.hip) against the host reader (fastx.cpp) and the oracle's reader
(`-m gpu`): the dictionaries built from the same files must be identical whichever side strips the headers and line breaks."""
import os

import numpy as np
import pytest

import ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def _dicts(E, files, k, host_parse, threads=4):
    if host_parse:
        os.environ["SKX_KNOBS"] = "host_parse=1"
    else:
        os.environ.pop("SKX_KNOBS", None)
    try:
        ds = E.DictSet.from_files([(f, None) for f in files], k, True, threads=threads)
        return [ds.export(i) for i in range(len(files))]
    finally:
        os.environ.pop("SKX_KNOBS", None)


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x[0]["lo"], y[0]["lo"]) and np.array_equal(x[0]["hi"], y[0]["hi"]) and np.array_equal(x[1], y[1])
                                    for x, y in zip(a, b))


def _rand_seq(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()


def _wrap(seq, width, eol=b"\n"):
    return eol.join(seq[o:o + width] for o in range(0, len(seq), width)) + (eol if seq else b"")


def test_device_parse_equals_host_and_oracle_on_ordinary_files(E, tmp_path):
    rng = np.random.default_rng(7)
    files = []
    for i, (width, n_rec, eol) in enumerate([(60, 30, b"\n"), (80, 3, b"\n"), (70, 12, b"\r\n"), (61, 200, b"\n"), (16384, 4, b"\n"), (1 << 30, 5, b"\n")]):
        p = tmp_path / f"s{i}.fa"
        with open(p, "wb") as f:
            for r in range(n_rec):
                seq = _rand_seq(rng, int(rng.integers(40, 90000)))
                if r % 3 == 1:
                    seq = seq[:100] + b"NNNNNnnnn" + seq[100:].lower()
                f.write(b">rec_%d some description\twith tabs" % r + eol + _wrap(seq, width, eol))
        files.append(str(p))
    for k in (31, 9):
        dev = _dicts(E, files, k, host_parse=False)
        host = _dicts(E, files, k, host_parse=True)
        assert _same(dev, host)
        for f, (gk, gb) in zip(files, dev):
            ok, ob = ora.Dict.from_files(k, f).export()
            assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gb, ob)


def test_device_parse_edge_cases_equal_host_reader(E, tmp_path):
    rng = np.random.default_rng(8)
    big = _rand_seq(rng, 70000)
    cases = {
        "no_trailing_newline": b">a\n" + _wrap(big[:5000], 60)[:-1],
        "blank_lines": b">a\n\n" + _wrap(big[:3000], 60) + b"\n\n>b\n\n" + _wrap(big[3000:9000], 50) + b"\n",
        "consecutive_headers": b">a\n>b\n>c\n" + _wrap(big[:4000], 60) + b">d\n>e\n" + _wrap(big[4000:8000], 60) + b">f\n",
        "gt_inside_line": b">a\n" + big[:300] + b">" + big[300:700] + b"\n" + big[700:1500] + b"\n>b\n" + _wrap(big[1500:4000], 60),
        "crlf_everywhere": b">a desc\r\n" + _wrap(big[:7000], 60, b"\r\n") + b">b\r\n" + _wrap(big[7000:9000], 60, b"\r\n"),
        "long_header": b">" + b"x" * 40000 + b"\n" + _wrap(big[:5000], 60) + b">" + b"y" * 16383 + b"\n" + _wrap(big[5000:12000], 60),
        "one_long_line": b">a\n" + big + b"\n>b\n" + big[::-1] + b"\n",
        "header_only_then_seq": b">only\n" + b">a\n" + _wrap(big[:2000], 60),
        "tile_boundary_headers": b">a\n" + big[:16381 - 3] + b"\n>b\n" + big[:16384] + b"\n>c\n" + big[100:16484 - 8] + b"\n>d\n" + _wrap(big[:3000], 61),
        "tiny": b">a\n" + big[:40] + b"\n",
    }
    names = sorted(cases)
    files = []
    for nm in names:
        p = tmp_path / f"{nm}.fa"
        p.write_bytes(cases[nm])
        files.append(str(p))
    for k in (7, 31):
        dev = _dicts(E, files, k, host_parse=False, threads=3)
        host = _dicts(E, files, k, host_parse=True, threads=3)
        for nm, d, h in zip(names, dev, host):
            assert _same([d], [h]), (nm, k, len(d[0]), len(h[0]))


def test_device_parse_errors_match_host_reader(E, tmp_path):
    empty = tmp_path / "e.fa"
    empty.write_bytes(b"")
    nohdr = tmp_path / "n.fa"
    nohdr.write_bytes(b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    onlyhdr = tmp_path / "h.fa"
    onlyhdr.write_bytes(b">nothing here\n")
    for f in (empty, nohdr, onlyhdr, tmp_path / "missing.fa"):
        errs = []
        for host in (False, True):
            with pytest.raises(E.EngineError) as ei:
                _dicts(E, [str(f)], 31, host_parse=host, threads=1)
            errs.append((ei.value.code, str(ei.value)))
        assert errs[0] == errs[1], errs


def test_bzip2_xz_zstd_inputs_equal_plain(E, tmp_path):
    """needletail (Cargo.toml:32, `compression`) reads bzip2 / xz / zstd besides gzip: the same dictionaries from every form of the same
    FASTA and FASTQ; a damaged file is "Invalid path/file", never parsed as text"""
    import bz2
    import ctypes
    import gzip
    import lzma
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = [acgt[rng.integers(0, 4, size=n)].tobytes() for n in (5000, 31, 700, 12000)]
    fa = b"".join(b">r%d\n" % i + b"\n".join(r[j:j + 60] for j in range(0, len(r), 60)) + b"\n" for i, r in enumerate(recs))
    fq = b"".join(b"@q%d\n" % i + r[:150] + b"\n+\n" + b"I" * len(r[:150]) + b"\n" for i, r in enumerate(recs * 30))

    def zstd(data):
        lib = ctypes.CDLL("libzstd.so.1")
        lib.ZSTD_compressBound.restype = ctypes.c_size_t; lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_compress.restype = ctypes.c_size_t
        lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        out = ctypes.create_string_buffer(lib.ZSTD_compressBound(len(data)))
        n = lib.ZSTD_compress(out, len(out), data, len(data), 3)
        return out.raw[:n]

    forms = {"plain": lambda d: d, "gz": gzip.compress, "bz2": lambda d: bz2.compress(d[:len(d) // 2]) + bz2.compress(d[len(d) // 2:]),      # two streams
             "xz": lzma.compress, "zst": zstd}
    for kind, data, q in (("fa", fa, None), ("fq", fq, E.qual(1, 0, E.QUAL_NOFILTER))):
        want = None
        for tag, f in forms.items():
            p = str(tmp_path / f"x_{kind}.{tag}")
            open(p, "wb").write(f(data))
            ds = E.DictSet.from_files([(p, None)], 31, True, q=q)
            got = ds.export(0)
            ds.free()
            if want is None:
                want = got
            else:
                assert np.array_equal(got[0]["lo"], want[0]["lo"]) and np.array_equal(got[1], want[1]), (kind, tag)
    bad = bz2.compress(fa)
    p = str(tmp_path / "broken.bz2")
    open(p, "wb").write(bad[:len(bad) // 2])
    with pytest.raises(E.EngineError) as ei:
        E.DictSet.from_files([(p, None)], 31, True)
    assert "Invalid path/file" in str(ei.value)
