"""`.fastq.gz` inflated on the device (skx_gzdev.hip) against zlib on the same bytes (`-m gpu`): the kernels around the logic that
tests/test_gz_device_logic.py checks on the host.  Status 0 means the text is zlib's byte for byte (member lengths and CRC-32s checked on the
device); anything else is refused, never read differently."""
import ctypes as C
import gzip
import os
import random
import struct
import zlib

import pytest

from test_gz_device_logic import fastq_text, gz, illumina_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def inflate(E, blob, hint=None, cap=None):
    lib = E.load_library()
    ctx = E.default_context()
    cap = cap if cap is not None else max(64, 70 * len(blob))
    text = C.create_string_buffer(cap)
    total, status, members = C.c_uint64(), C.c_uint32(), C.c_uint32()
    ms = (C.c_double * 2)()
    lib.skx_debug_gz_inflate.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.skx_debug_gz_inflate.restype = C.c_int
    rc = lib.skx_debug_gz_inflate(ctx.h, blob, len(blob), hint if hint is not None else 3 * len(blob), text, cap, C.byref(total), C.byref(status), C.byref(members), ms)
    assert rc == 0, lib.skx_last_error()
    return status.value, members.value, (text.raw[:total.value] if status.value == 0 else None), (ms[0], ms[1])


@pytest.mark.parametrize("level", [1, 6, 9])
def test_levels(E, level, monkeypatch):
    text = fastq_text(20000, level)
    for knobs in ("", "gz_chunk_kb=8,gz_group=3", "gz_chunk_kb=16,gz_group=1", "gz_verify=1", "gz_lane0=1"):
        monkeypatch.setenv("SKX_KNOBS", knobs)
        status, members, got, _ = inflate(E, gz(text, level), hint=len(text))
        assert status == 0 and members == 1 and got == text, (level, knobs, status)


def test_sequencer_style_reads(E, monkeypatch):
    monkeypatch.setenv("SKX_KNOBS", "")
    for level in (1, 5, 9):
        text = illumina_text(40000, level)
        status, members, got, _ = inflate(E, gz(text, level), hint=len(text))
        assert status == 0 and members == 1 and got == text, level


def test_block_kinds_members_and_headers(E, monkeypatch):
    monkeypatch.setenv("SKX_KNOBS", "gz_chunk_kb=16,gz_group=4,gz_walk_kb=65536")
    text = fastq_text(5000, 3)
    pieces = [text[i:i + 65280] for i in range(0, len(text), 65280)]

    def bgzf_block(p):
        body = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = body.compress(p) + body.flush()
        head = b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(raw) + 25)
        return head + raw + struct.pack("<II", zlib.crc32(p), len(p))
    bg = b"".join(bgzf_block(p) for p in pieces) + bgzf_block(b"")
    named = b"\x1f\x8b\x08\x18" + b"\0" * 4 + b"\0\x03" + b"reads_1.fastq\0" + b"a comment\0"
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    named += c.compress(text) + c.flush() + struct.pack("<II", zlib.crc32(text), len(text))
    joined = gz(text[:300_000], 1) + gz(text[300_000:700_000], 9) + gz(text[700_000:], 6)
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    flushed = b""
    for i in range(0, len(text), 100_000):
        flushed += c.compress(text[i:i + 100_000]) + c.flush(zlib.Z_FULL_FLUSH if (i // 100_000) % 2 else zlib.Z_SYNC_FLUSH)
    flushed += c.flush()
    cases = [(bg, len(pieces) + 1), (named, 1), (joined, 3), (flushed, 1), (gz(text, 0), 1), (gz(text, 6, zlib.Z_FIXED), 1), (gz(text, 6, zlib.Z_HUFFMAN_ONLY), 1),
             (gz(text, 6, zlib.Z_RLE), 1), (gz(text, 6, zlib.Z_DEFAULT_STRATEGY, 1), 1)]
    for blob, nm in cases:
        status, members, got, _ = inflate(E, blob)
        assert status == 0 and got == text and members == nm, (status, members, nm)
    rng = random.Random(5)
    for small in (b"", b"A", b"@r\nACGT\n+\nFFFF\n", bytes(rng.getrandbits(8) for _ in range(200_000))):
        status, members, got, _ = inflate(E, gz(small, 6))
        assert status == 0 and got == small
    # a long stretch the finder has no place to start in (stored blocks; literals that are not text) is not walked by one wavefront: refused
    monkeypatch.setenv("SKX_KNOBS", "gz_chunk_kb=16")
    assert inflate(E, gz(text, 0))[0] == 4
    assert inflate(E, gz(bytes(rng.getrandbits(8) for _ in range(3_000_000)), 6))[0] in (0, 4)


def test_refusals(E, monkeypatch):
    monkeypatch.setenv("SKX_KNOBS", "gz_chunk_kb=16,gz_group=4")
    text = fastq_text(4000, 9)
    blob = gz(text, 6)
    rng = random.Random(1)
    cases = [blob[:len(blob) // 2], blob[:-1], blob[:-8], blob[:-4] + b"\0\0\0\0", blob[:-8] + b"\0\0\0\0" + blob[-4:], blob + b"\0" * 7, blob + b"\x1f\x8b", blob + b"junk"]
    for _ in range(25):
        b = bytearray(blob)
        p = rng.randrange(12, len(b) - 8)
        b[p] ^= 1 << rng.randrange(8)
        cases.append(bytes(b))
    refused = 0
    for blob2 in cases:
        status, members, got, _ = inflate(E, blob2, hint=len(text))
        try:
            ref = gzip.decompress(blob2)
        except Exception:
            ref = None
        if status == 0:
            assert ref is not None and got == ref
        else:
            refused += 1
    assert refused >= 30
    # a stretch that deflates beyond the symbol area is refused (status 2), and decoded when the area is sized for it
    rep = gz(b"G" * 3_000_000 + text[:50_000], 9)
    monkeypatch.setenv("SKX_KNOBS", "gz_chunk_kb=4,gz_ratio=8,gz_walk_kb=65536")
    assert inflate(E, rep, cap=4_000_000)[0] == 2
    monkeypatch.setenv("SKX_KNOBS", "gz_chunk_kb=4,gz_ratio=2048,gz_walk_kb=65536")
    status, members, got, _ = inflate(E, rep, cap=4_000_000)
    assert status == 0 and got == b"G" * 3_000_000 + text[:50_000]
