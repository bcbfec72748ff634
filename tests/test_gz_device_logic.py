"""The device inflater's logic (ska.rust_amd/csrc/gz_device.h: block finder, symbolic chunk decode, window maps, member lengths and CRCs) run
on the HOST by tools/gzdev_host_check.cpp in the kernels' order and layout, against zlib's text for gzip files of every kind the reader threads'
inflater is tested with (tests/test_gz_reader.py): compression levels and strategies, stored and fixed blocks, several members, bgzip-like
blocks, flush points, header fields, damaged and truncated files.  The rule: status 0 means the text is zlib's, byte for byte; anything the
decoder does not vouch for is a non-zero status (the engine then sends the sample through the reader threads' inflater)."""
import gzip, os, random, struct, subprocess, zlib
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def check(tmp_path_factory):
    d = tmp_path_factory.mktemp("gzdev")
    exe = str(d / "gzdev_host_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tools", "gzdev_host_check.cpp")], check=True)

    def run(blob, chunk=65536, ratio=8, group=32):
        src, out = str(d / "in.gz"), str(d / "out.txt")
        open(src, "wb").write(blob)
        if os.path.exists(out):
            os.unlink(out)
        r = subprocess.run([exe, src, out, str(chunk), str(ratio), str(group)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        status, total, members, chunks, synced = (int(x) for x in r.stdout.split())
        text = open(out, "rb").read() if status == 0 else None
        return status, total, members, chunks, synced, text
    return run


def fastq_text(n_reads, seed, read_len=150):
    rng = np.random.default_rng(seed)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=200_000)
    out = []
    for i in range(n_reads):
        p = int(rng.integers(0, len(genome) - read_len))
        q = rng.choice(np.frombuffer(b"#,5:AFFFFF", dtype=np.uint8), size=read_len)
        out.append(b"@run7:%d:%d/1\n" % (seed, i) + genome[p:p + read_len].tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    return b"".join(out)


def illumina_text(n_reads, seed):
    """reads the way a sequencer's software writes them: long headers that differ in a few digits, lengths that vary (trimmed reads), the full
    quality alphabet falling off along the read, N calls, a comment behind the '+' now and then"""
    rng = np.random.default_rng(seed)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300_000)
    out = []
    for i in range(n_reads):
        ln = int(rng.integers(35, 152))
        p = int(rng.integers(0, len(genome) - ln))
        seq = genome[p:p + ln].copy()
        seq[rng.random(ln) < 0.003] = ord("N")
        q = np.clip(40 - (np.arange(ln) // 12) + rng.integers(-8, 2, size=ln), 2, 41).astype(np.uint8) + 33
        head = b"@A00%d:%d:HXY%dDSXX:%d:%d:%d:%d %d:N:0:ACGTTGCA+TTGACCGA" % (seed, 17 + seed, seed, 1 + i % 4, 1101 + i // 5000, int(rng.integers(1000, 32000)), int(rng.integers(1000, 37000)), 1 + i % 2)
        out.append(head + b"\n" + seq.tobytes() + b"\n+" + (head[1:] if i % 97 == 0 else b"") + b"\n" + q.tobytes() + b"\n")
    return b"".join(out)


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return c.compress(data) + c.flush()


@pytest.mark.parametrize("level", [1, 4, 6, 9])
@pytest.mark.parametrize("chunk,group", [(65536, 32), (8192, 3), (16384, 1)])
def test_levels_and_chunkings(check, level, chunk, group):
    text = fastq_text(6000, level)
    status, total, members, chunks, synced, got = check(gz(text, level), chunk, 8, group)
    assert status == 0 and got == text and members == 1
    assert chunks < 3 or synced > 0                          # (the finder does find blocks: the chunks are not all walked by the first one)


@pytest.mark.parametrize("level", [1, 5, 9])
def test_sequencer_style_reads(check, level):
    text = illumina_text(12000, level)
    status, total, members, chunks, synced, got = check(gz(text, level), 16384, 12, 64)
    assert status == 0 and got == text and synced > chunks // 4


def test_strategies_stored_fixed_and_memlevel(check):
    text = fastq_text(3000, 11)
    for blob in (gz(text, 0), gz(text, 6, zlib.Z_FIXED), gz(text, 6, zlib.Z_HUFFMAN_ONLY), gz(text, 6, zlib.Z_RLE), gz(text, 9, zlib.Z_FILTERED),
                 gz(text, 6, zlib.Z_DEFAULT_STRATEGY, 1), gz(text, 1, zlib.Z_DEFAULT_STRATEGY, 9)):
        status, total, members, chunks, synced, got = check(blob, 16384, 10, 4)
        assert status == 0 and got == text


def test_members_bgzip_like_and_flush_points(check):
    text = fastq_text(5000, 3)
    # members joined (cat a.gz b.gz), 64 KB members with an extra field the way bgzip writes them, and one with a name and a comment
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
    for blob, nm in ((bg, len(pieces) + 1), (named, 1), (joined, 3), (flushed, 1)):
        for chunk, group in ((65536, 32), (8192, 2)):
            status, total, members, chunks, synced, got = check(blob, chunk, 12, group)
            assert status == 0 and got == text and members == nm, (status, members, nm)


def test_small_empty_and_binary(check):
    rng = random.Random(5)
    for text in (b"", b"A", b"@r\nACGT\n+\nFFFF\n", bytes(rng.getrandbits(8) for _ in range(200_000)), b"ACGT" * 10):
        status, total, members, chunks, synced, got = check(gz(text, 6), 4096, 16, 2)
        assert status == 0 and got == text


def test_long_matches_ratio_beyond_the_symbol_area(check):
    text = b"G" * 3_000_000 + fastq_text(500, 1)
    blob = gz(text, 9)
    status, *_ = check(blob, 4096, 8, 4)
    assert status == 2                                        # (E_OVERFLOW: the sample goes to the reader threads' inflater)
    status, total, members, chunks, synced, got = check(blob, 4096, 64 * 32, 4)
    assert status == 0 and got == text


def test_damaged_and_truncated_files_never_pass(check):
    text = fastq_text(4000, 9)
    blob = gz(text, 6)
    rng = random.Random(1)
    wrong = 0
    cases = [blob[:len(blob) // 2], blob[:-1], blob[:-8], blob[:-4] + b"\0\0\0\0", blob[:-8] + b"\0\0\0\0" + blob[-4:], blob + b"\0" * 7, blob + b"\x1f\x8b", blob + b"junk"]
    for _ in range(40):
        b = bytearray(blob)
        p = rng.randrange(12, len(b) - 8)
        b[p] ^= 1 << rng.randrange(8)
        cases.append(bytes(b))
    for blob2 in cases:
        status, total, members, chunks, synced, got = check(blob2, 16384, 10, 4)
        try:
            ref = gzip.decompress(blob2)
        except Exception:
            ref = None
        if status == 0:
            assert ref is not None and got == ref              # passed: then it is what zlib makes of the same bytes
        else:
            wrong += 1
    assert wrong >= 40                                        # (nearly every damaged file is refused; a flipped bit zlib also accepts is impossible: the CRC)
