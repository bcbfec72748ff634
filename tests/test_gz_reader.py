"""The reader threads' own inflater (csrc/gz_inflate.cpp) against Python's zlib, on the CPU: every kind of DEFLATE block, gzip header
fields, several members, trailing bytes, and the damaged / truncated files that must be errors (needletail through flate2 refuses
them: ska_dict.rs:131-153) -- through skx_read_records, both the one-shot reader and the line-by-line FASTQ reader of the read-set
pipeline."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

import skx_engine as E


def _fastq(n_reads, read_len, seed, qual_mode="profile", crlf=False, n_rate=0.001):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, 200_000, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    prof = bytes(33 + max(2, 40 - (i * i) // 600) for i in range(read_len))
    out = []
    eol = b"\r\n" if crlf else b"\n"
    for r in range(n_reads):
        p = int(rng.integers(0, len(genome) - read_len))
        s = lut[genome[p:p + read_len]].copy()
        if n_rate:
            s[rng.random(read_len) < n_rate] = ord("N")
        if qual_mode == "profile":
            q = prof
        else:
            q = bytes(rng.integers(33, 74, read_len, dtype=np.uint8))
        out.append(b"@read_%d/1" % r + eol + s.tobytes() + eol + b"+" + eol + q + eol)
    return b"".join(out)


def _expect_fastq(text):
    lines = text.replace(b"\r\n", b"\n").split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    seq = b"".join(l + b"\n" for l in lines[1::4])
    qual = b"".join(l + b"\n" for l in lines[3::4])
    return seq, qual


def _gz_member(data, level=6, fname=None, comment=None, extra=None, hcrc=False, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15):
    flg = (8 if fname else 0) | (16 if comment else 0) | (4 if extra else 0) | (2 if hcrc else 0)
    h = b"\x1f\x8b\x08" + bytes([flg]) + struct.pack("<IBB", 0, 0, 255)
    if extra:
        h += struct.pack("<H", len(extra)) + extra
    if fname:
        h += fname + b"\0"
    if comment:
        h += comment + b"\0"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    c = zlib.compressobj(level, zlib.DEFLATED, -wbits, 9, strategy)
    body = c.compress(data) + c.flush()
    return h + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def _both(path, expect_seq, expect_qual):
    for streaming in (False, True):
        s, q = E.read_records(path, streaming=streaming)
        assert s == expect_seq, (streaming, len(s), len(expect_seq))
        assert q == expect_qual, streaming


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_gz_reader_block_kinds(tmp_path, level, strategy):
    # stored blocks (level 0), fixed codes, dynamic codes, literals only, distance-1 runs
    text = _fastq(6000, 151, seed=level * 10 + strategy)
    p = tmp_path / "a.fastq.gz"
    p.write_bytes(_gz_member(text, level=level, strategy=strategy))
    _both(str(p), *_expect_fastq(text))


def test_gz_reader_random_qualities_and_crlf(tmp_path):
    text = _fastq(5000, 100, seed=3, qual_mode="random", crlf=True)
    p = tmp_path / "a.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    _both(str(p), *_expect_fastq(text))


def test_gz_reader_header_fields_members_and_trailing_bytes(tmp_path):
    a, b, c = _fastq(3000, 151, 1), _fastq(10, 50, 2), _fastq(4000, 75, 3, qual_mode="random")
    blob = (_gz_member(a, fname=b"reads_1.fastq", comment=b"made by a test", extra=b"BC\x02\x00\x00\x10", hcrc=True)
            + _gz_member(b, level=1) + _gz_member(b"") + _gz_member(c, level=9, wbits=9) + b"\0" * 37)       # bgzip-like fields, an empty member, a small window, padding
    p = tmp_path / "a.fastq.gz"
    p.write_bytes(blob)
    _both(str(p), *_expect_fastq(a + b + c))


def test_gz_reader_text_larger_than_every_buffer(tmp_path):
    # > the inflater's 1 MB window and 1 MB input buffer, several times; a last line without a line end
    text = _fastq(60_000, 151, 9, qual_mode="random")[:-1]
    assert len(text) > 16 << 20
    p = tmp_path / "big.fastq.gz"
    p.write_bytes(gzip.compress(text, 1))
    _both(str(p), *_expect_fastq(text))
    # and through the one-shot reader as FASTA (one long record: no line end for megabytes)
    fa = b">contig\n" + text.replace(b"\n", b"") + b"\n"
    p2 = tmp_path / "long.fa.gz"
    p2.write_bytes(gzip.compress(fa, 6))
    s, q = E.read_records(str(p2))
    assert q is None and s == text.replace(b"\n", b"") + b"\n"


def test_gz_reader_a_line_longer_than_the_kept_window(tmp_path):
    # FASTQ whose lines are longer than what the inflater keeps for the caller (the caller puts them aside)
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 700_000)].tobytes()
    qual = bytes(rng.integers(33, 74, 700_000, dtype=np.uint8))
    text = b"@long\n" + seq + b"\n+\n" + qual + b"\n" + _fastq(100, 151, 1)
    p = tmp_path / "l.fastq.gz"
    p.write_bytes(gzip.compress(text, 6))
    _both(str(p), *_expect_fastq(text))


def test_gz_reader_agrees_with_zlib_reader_on_fixtures(tmp_path):
    # the reference's own gzip fixtures, and the knob that brings zlib's gzread back: same records
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    found = []
    for dp, _, files in os.walk(root):
        found += [os.path.join(dp, f) for f in files if f.endswith(".gz")]
    assert found
    for f in sorted(found):
        raw = gzip.decompress(open(f, "rb").read())
        s, q = E.read_records(f)
        if raw[:1] == b"@":
            assert (s, q) == _expect_fastq(raw), f
            assert E.read_records(f, streaming=True) == (s, q), f


def test_gz_reader_refuses_damaged_files(tmp_path):
    text = _fastq(20_000, 151, 11, qual_mode="random")
    good = gzip.compress(text, 6)
    p = tmp_path / "x.fastq.gz"
    rnd = random.Random(1)

    def refused(blob):
        p.write_bytes(blob)
        for streaming in (False, True):
            with pytest.raises(E.EngineError) as ei:
                E.read_records(str(p), streaming=streaming)
            assert ei.value.code == E.EIO, (streaming, ei.value)

    refused(good[:len(good) // 2])                      # truncated in the middle of a block
    refused(good[:-8])                                  # trailer missing
    refused(good[:-3])                                  # trailer cut
    refused(good[:-8] + struct.pack("<II", (zlib.crc32(text) ^ 1) & 0xFFFFFFFF, len(text)))      # wrong CRC
    refused(good[:-4] + struct.pack("<I", len(text) + 1))                                        # wrong length
    refused(good[:10] + b"\x07" + good[11:])            # block type 3
    refused(b"\x1f\x8b\x08\xe0" + good[4:])             # reserved header flags
    refused(b"\x1f\x8b\x07" + good[3:])                 # not DEFLATE
    bad_dist = b"\x1f\x8b\x08\x00" + b"\0" * 6 + zlib.compressobj(6, zlib.DEFLATED, -15).compress(b"") + bytes([0x73, 0x04, 0x91, 0x00]) + b"\0" * 8
    refused(bad_dist)
    flips = 0
    for _ in range(200):                                # a flipped bit anywhere: an error or, if the text survives (header time stamp), the same text
        i = rnd.randrange(len(good))
        blob = bytearray(good)
        blob[i] ^= 1 << rnd.randrange(8)
        p.write_bytes(bytes(blob))
        try:
            s, q = E.read_records(str(p), streaming=True)
        except E.EngineError as e:
            assert e.code == E.EIO
            flips += 1
            continue
        assert (s, q) == _expect_fastq(text), i
    assert flips > 150


def test_gz_reader_member_checksums_at_every_length(tmp_path):
    # one member per length 0..600 (and a few around the folded CRC's block sizes): each member's CRC-32 is checked over exactly that
    # many bytes, short ones by the table form, long ones by the carry-less-multiplication form, at whatever alignment the window gives
    rng = np.random.default_rng(2)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    members, expect = [], []
    for i, n in enumerate(list(range(0, 600)) + [1023, 1024, 1025, 4096 + 15, 65_536 + 1, 300_001]):
        body = lut[rng.integers(0, 4, n)].tobytes()
        rec = b">r%d\n" % i + body + b"\n"
        members.append(_gz_member(rec, level=(0, 1, 6)[i % 3]))
        expect.append(body + b"\n")
    p = tmp_path / "m.fa.gz"
    p.write_bytes(b"".join(members))
    s, q = E.read_records(str(p))
    assert q is None and s == b"".join(expect)
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import skx_engine as E; s, q = E.read_records(%r); "
            "import hashlib; print(hashlib.sha1(s).hexdigest())" % (os.path.dirname(E.__file__), str(p)))
    import hashlib
    for knobs in ("no_clmul", "zlib_reader"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SKX_KNOBS=knobs), capture_output=True, text=True, check=True).stdout.strip()
        assert out == hashlib.sha1(b"".join(expect)).hexdigest(), knobs


def test_gz_reader_random_streams(tmp_path):
    """Seeded streams of every zlib strategy (default, filtered, Huffman only, RLE, fixed codes), level 0-9, window 2^9-2^15, memLevel 1-9, over
    five kinds of bytes (uniform, four letters, long runs, short periods, re-used stretches), cut into up to three members: the records that
    come out are the bytes that went in."""
    rnd = random.Random(7)
    allowed = np.array([b for b in range(32, 127) if b != ord(">")], dtype=np.uint8)
    p = str(tmp_path / "f.fa.gz")

    def payload(kind, n):
        rng = np.random.default_rng(rnd.randrange(1 << 30))
        if kind == 0:
            return allowed[rng.integers(0, len(allowed), n)].tobytes()
        if kind == 1:
            return allowed[rng.integers(0, 4, n)].tobytes()
        if kind == 2:
            out = bytearray()
            while len(out) < n:
                out += bytes([int(allowed[rng.integers(0, len(allowed))])]) * int(rng.integers(1, 600))
            return bytes(out[:n])
        if kind == 3:
            unit = allowed[rng.integers(0, len(allowed), int(rng.integers(1, 40)))].tobytes()
            return (unit * (n // len(unit) + 1))[:n]
        base = allowed[rng.integers(0, 8, 5000)].tobytes()
        out = bytearray()
        while len(out) < n:
            o = int(rng.integers(0, 4900))
            out += base[o:o + int(rng.integers(3, 300))]
        return bytes(out[:n])

    for it in range(80):
        kind = rnd.randrange(5)
        n = rnd.choice([0, 1, 5, 100, 70_000, 300_000, 1_100_000]) + rnd.randrange(50)
        body = payload(kind, n)
        text = b">r\n" + body + b"\n"
        cuts = sorted(rnd.sample(range(1, len(text)), min(len(text) - 1, rnd.randrange(0, 3))))
        parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
        blob = b"".join(_gz_member(pt, level=rnd.randrange(0, 10), wbits=rnd.randrange(9, 16),
                                   strategy=rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])) for pt in parts)
        with open(p, "wb") as f:
            f.write(blob)
        s, q = E.read_records(p)
        assert q is None and s == body + b"\n", (it, kind, n)


def test_gz_reader_file_that_ends_inside_a_header(tmp_path):
    """A file cut inside a dynamic block's code-length header, or inside the gzip header of a further member, is truncated -- an error,
    never a short input; only bytes that do not begin like a member are ignored behind the last one (round 5's advisor findings: the
    code-length loops check the input's end at every refill, and `1f 8b ...` cut short is not trailing garbage)."""
    text = _fastq(4000, 151, 21, qual_mode="random")
    good = gzip.compress(text, 6)
    p = tmp_path / "t.fastq.gz"

    def refused(blob):
        p.write_bytes(blob)
        for streaming in (False, True):
            with pytest.raises(E.EngineError) as ei:
                E.read_records(str(p), streaming=streaming)
            assert ei.value.code == E.EIO

    for cut in range(11, 140):                          # the first block's header: HLIT / HDIST / HCLEN, the code-length code, the lengths
        refused(good[:cut])
    # the same with the cut just under the input buffer's size, where the loads behind the end would leave the buffer: a stored member
    # of about 1 MB first, so that the header lies at the buffer's end
    pad_text = _fastq(3500, 151, 22, qual_mode="random")
    pad = _gz_member(pad_text, level=0)
    for want in ((1 << 20) - 3, (1 << 20) - 40, (1 << 20) + 64 - 20):
        fill = want - len(pad) - 12
        if fill < 0:
            continue
        blob = pad + good[:12 + fill] if fill < 100 else None
        if blob is None:                                # stretch the stored member instead
            extra = _fastq((fill // 310) + 1, 151, 23, qual_mode="random")
            pad2 = _gz_member(extra[:max(0, want - len(pad) - 60 - 18)], level=0)
            blob = pad + pad2
            blob = blob + good[:max(12, want - len(blob))]
        refused(blob)
    for k in range(1, 10):                              # a second member's header cut after k bytes
        refused(good + good[:k])
    p.write_bytes(good + b"\x00\x01\x02\x03\x04")       # not a member: ignored, as gzip ignores it
    _both(str(p), *_expect_fastq(text))
    p.write_bytes(good + b"\x1e")
    _both(str(p), *_expect_fastq(text))
