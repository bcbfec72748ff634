"""FASTQ text -> bit planes on the device (skx_fastq.hip: line ends, four-line framing, the record checks, 5-bit planes) against the host
reader (fastx.cpp stream_fastq_file, the needletail record iterator's stand-in: ska_dict.rs:131-153,356-366) on the same bytes (`-m gpu`).
What the device accepts must be exactly the records the host reader delivers; what it calls irregular must be either something the host
reader accepts by its slower rules (blank lines between records) or refuses -- never a text the device silently reads differently.  Through
`ska build`, files of every such kind give the .skf of the one-shot form or its error."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")


@pytest.fixture(scope="module")
def E():
    import skx_engine as eng
    eng.load_library()
    eng.default_context()
    return eng


def _frame(E, text, junction=0, min_qual=20):
    lib = E.load_library()
    ctx = E.default_context()
    n = len(text) // 2 + 64
    seq, qual = C.create_string_buffer(n), C.create_string_buffer(n)
    pos, irr = C.c_uint64(), C.c_int()
    lib.skx_debug_fastq_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    lib.skx_debug_fastq_frame.restype = C.c_int
    rc = lib.skx_debug_fastq_frame(ctx.h, text, len(text), junction, min_qual, seq, qual, C.byref(pos), C.byref(irr))
    assert rc == 0, lib.skx_last_error()
    return bool(irr.value), seq.raw[:pos.value], qual.raw[:pos.value]


_CANON = np.zeros(256, np.uint8)
for _c in range(256):
    _CANON[_c] = ord("N") if (_c & 15) == 14 else b"ACTG"[(_c >> 1) & 3]
_CANON[10] = 10


def _canon(seq, qual, min_qual):
    s = np.frombuffer(seq, np.uint8)
    q = np.frombuffer(qual, np.uint8)
    nl = s == 10
    cs = _CANON[s]
    cq = np.where(((q.astype(np.int32) - 33) & 255) <= min_qual, ord("!"), ord(" ")).astype(np.uint8)
    cq[nl] = 10
    return cs.tobytes(), cq.tobytes()


def _host(E, tmp_path, text, text2=None):
    p = os.path.join(str(tmp_path), "h.fastq")
    open(p, "wb").write(text)
    p2 = None
    if text2 is not None:
        p2 = os.path.join(str(tmp_path), "h2.fastq")
        open(p2, "wb").write(text2)
    return E.read_records(p, p2, streaming=True)


def _records(rng, n, lens, eol=b"\n", header=lambda i: b"@read_%d some text" % i, plus=lambda i: b"+"):
    out = []
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNnRYKMSWacgt.>-*", np.uint8)
    for i in range(n):
        L = int(lens[rng.integers(0, len(lens))])
        s = alphabet[rng.integers(0, len(alphabet), size=L)].tobytes()
        q = (33 + rng.integers(0, 42, size=L)).astype(np.uint8).tobytes()
        out.append(header(i) + eol + s + eol + plus(i) + eol + q + eol)
    return b"".join(out)


def test_device_framing_equals_host_reader_on_regular_text(E, tmp_path):
    rng = np.random.default_rng(11)
    lens_all = np.array([0, 1, 2, 30, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 150, 151, 250, 1000])
    cases = {
        "ragged": _records(rng, 20_000, lens_all),
        "uniform_150": _records(rng, 30_000, np.array([150])),
        "crlf": _records(rng, 5_000, lens_all, eol=b"\r\n"),
        "plus_repeats_name": _records(rng, 5_000, lens_all, plus=lambda i: b"+read_%d some text" % i),
        "at_and_plus_in_quality": b"".join(b"@r\n" + b"ACGT" * 10 + b"\n+\n" + (b"@+" * 20) + b"\n" for _ in range(3000)),
        "no_final_newline": _records(rng, 3_000, lens_all)[:-1] if True else b"",
        "one_record": b"@r\nACGTN\n+\nIIII!\n",
        "empty_reads_only": b"@r\n\n+\n\n" * 5000,
        "long_lines": _records(rng, 40, np.array([20_000, 70_000, 16_384, 16_383, 4096, 4095])),
        "tile_edges": b"".join(b"@" + b"h" * int(h) + b"\n" + b"ACGT" * 25 + b"\n+\n" + b"I" * 100 + b"\n" for h in rng.integers(1, 400, size=4000)),
    }
    for name, text in cases.items():
        t = text if text.endswith(b"\n") else text + b"\n"                  # (the pipeline's reader puts the missing '\n' behind a file)
        for mq in (20, 0, 41, 255):
            irr, ds, dq = _frame(E, t, 0, mq)
            assert not irr, name
            hs, hq = _host(E, tmp_path, text)
            assert (ds, dq) == _canon(hs, hq, mq), (name, mq)
            if name not in ("ragged", "crlf"):
                break
    # two files: file 2 behind file 1, the junction where a record begins
    a, b = _records(rng, 4000, lens_all), _records(rng, 3000, lens_all, eol=b"\r\n")
    irr, ds, dq = _frame(E, a + b, len(a), 20)
    hs, hq = _host(E, tmp_path, a, b)
    assert not irr and (ds, dq) == _canon(hs, hq, 20)


def test_device_framing_calls_everything_else_irregular(E, tmp_path):
    rng = np.random.default_rng(12)
    lens = np.array([10, 50, 100])
    good = _records(rng, 2000, lens)
    recs = good.split(b"\n")
    accepted_by_host = {
        "blank_line_between_records": b"\n".join(recs[:400]) + b"\n\n" + b"\n".join(recs[400:]),
        "blank_lines_at_the_end": good + b"\n\n",
        "cr_only_line_between_records": b"\n".join(recs[:800]) + b"\n\r\n" + b"\n".join(recs[800:]),
    }
    refused_by_host = {
        "truncated_in_quality": good[:len(good) - 30],
        "truncated_after_plus": b"\n".join(recs[:4 * 100 + 3]) + b"\n",
        "three_line_record": b"\n".join(recs[:401] + recs[402:]),
        "missing_at": good.replace(b"@read_700 ", b"read_700 ", 1),
        "missing_plus": b"\n".join(recs[:4 * 50 + 2] + [b"-"] + recs[4 * 50 + 3:]),
        "lengths_differ": b"\n".join(recs[:4 * 60 + 3] + [recs[4 * 60 + 3] + b"I"] + recs[4 * 60 + 4:]),
        "fasta": b">a\nACGT\n>b\nACGT\n",
        "blank_first_line": b"\n" + good,                                  # (the streaming reader wants '@' as the file's first byte)
        "blank_line_inside_record": b"\n".join(recs[:4 * 70 + 1]) + b"\n\n" + b"\n".join(recs[4 * 70 + 1:]),
    }
    for name, text in accepted_by_host.items():
        t = text if text.endswith(b"\n") else text + b"\n"
        irr, _, _ = _frame(E, t)
        assert irr, name
        hs, hq = _host(E, tmp_path, text)                                   # the host reader's slower rules take these
        assert hs.count(b"\n") == 2000, name
    for name, text in refused_by_host.items():
        t = text if text.endswith(b"\n") else text + b"\n"
        irr, _, _ = _frame(E, t)
        assert irr, name
        with pytest.raises(E.EngineError):
            _host(E, tmp_path, text)
    # file 1 ends inside a record, file 2 supplies the rest: line counts add up, the junction does not
    half = b"\n".join(recs[:4 * 300 + 2]) + b"\n"
    rest = b"\n".join(recs[4 * 300 + 2:])
    irr, _, _ = _frame(E, half + rest, len(half))
    assert irr
    irr, _, _ = _frame(E, half + rest, 0)                                    # (as one file the same bytes are regular)
    assert not irr


def test_ska_build_on_unusual_fastq_takes_the_host_readers_verdict(E, tmp_path):
    """Through the executable with every sample sent raw: files the device calls irregular give the one-shot form's .skf (blank lines, a last
    line without its end) or its error (a record cut short) -- and so does the pipeline left to choose, and with the readers packing."""
    import synth
    wd = str(tmp_path)
    anc = synth.ancestor(40_000, seed=9)
    n = 5
    pairs = [synth.write_read_pair(anc, i, n, os.path.join(wd, f"r{i}"), read_len=100, coverage=20.0, seed=9) for i in range(n)]
    t = open(pairs[1][0], "rb").read()
    open(pairs[1][0], "wb").write(t[:-1])                                    # no final newline: regular once the reader has put it there
    t = open(pairs[2][1], "rb").read()
    cut = t.index(b"\n@r\n", len(t) // 2) + 1
    open(pairs[2][1], "wb").write(t[:cut] + b"\n" + t[cut:] + b"\n\n")       # blank lines between records and at the end
    t = open(pairs[3][0], "rb").read()
    open(pairs[3][0], "wb").write(t.replace(b"\n", b"\r\n"))                  # CRLF
    with open(os.path.join(wd, "list.txt"), "w") as f:
        for i, (a, b) in enumerate(pairs):
            f.write(f"r{i}\t{a}\t{b}\n")
    outs = {}
    for tag, knobs in (("raw", "reads_raw=2"), ("auto", ""), ("packed", "reads_raw=1"), ("oneshot", "no_reads_pipeline=1")):
        r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", tag, "-k", "31", "--min-count", "2", "--threads", "3"], cwd=wd, capture_output=True, timeout=300,
                           env=dict(os.environ, SKX_KNOBS=knobs, SKX_PHASES=os.path.join(wd, tag + ".json")))
        assert r.returncode == 0, r.stderr[-600:]
        outs[tag] = open(os.path.join(wd, tag + ".skf"), "rb").read()
    assert outs["raw"] == outs["oneshot"] and outs["auto"] == outs["oneshot"] and outs["packed"] == outs["oneshot"]
    import json
    ph = json.load(open(os.path.join(wd, "raw.json")))
    assert ph.get("build.reads_samples_sent_raw", 0) >= 1 and ph.get("build.reads_samples_irregular", 0) >= 1      # the case this test is about did arise
    t = open(pairs[4][1], "rb").read()
    open(pairs[4][1], "wb").write(t[:len(t) // 2 - 11])                       # a record cut in the middle
    for knobs in ("reads_raw=2", "", "reads_raw=1", "no_reads_pipeline=1"):
        r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", "bad", "-k", "31", "--min-count", "2", "--threads", "3"], cwd=wd, capture_output=True, timeout=300,
                           env=dict(os.environ, SKX_KNOBS=knobs))
        assert r.returncode != 0 and b"Invalid FASTA/Q record" in r.stderr, (knobs, r.stderr[-400:])


def test_device_framing_fuzz_against_the_host_reader(E, tmp_path):
    """Seeded mutations of small FASTQ texts -- bytes flipped to '\\n', '\\r', '@', '+' or anything, line ends removed or doubled, pieces cut out or
    repeated, the text cut short: whatever the device accepts as regular is what the host reader delivers, record for record; whatever the host
    reader refuses the device calls irregular (the other direction -- texts the host reader accepts by its slower rules -- is the device's
    right to pass on).  900 texts of 40 to 4 000 bytes, three quality thresholds."""
    rng = np.random.default_rng(2026)
    lens = np.array([0, 1, 2, 5, 17, 31, 32, 33, 64, 65, 100, 150])
    regular = irregular = refused = 0
    for it in range(900):
        base = bytearray(_records(rng, int(rng.integers(1, 14)), lens, eol=b"\r\n" if it % 7 == 0 else b"\n",
                                  header=lambda i: b"@r%d" % i, plus=(lambda i: b"+") if it % 3 else (lambda i: b"+r%d" % i)))
        for _ in range(int(rng.integers(0, 4))):
            if not base:
                break
            kind = int(rng.integers(0, 7))
            pos = int(rng.integers(0, len(base)))
            if kind == 0:
                base[pos] = int(rng.choice(np.frombuffer(b"\n\r@+ACGTN!I~", np.uint8)))
            elif kind == 1:
                base[pos] = int(rng.integers(1, 256))
            elif kind == 2:
                nl = [i for i, c in enumerate(base) if c == 10]
                if nl:
                    del base[nl[int(rng.integers(0, len(nl)))]]
            elif kind == 3:
                base[pos:pos] = b"\n"
            elif kind == 4:
                end = min(len(base), pos + int(rng.integers(1, 60)))
                del base[pos:end]
            elif kind == 5:
                end = min(len(base), pos + int(rng.integers(1, 60)))
                base[pos:pos] = base[pos:end]
            else:
                del base[pos:]
        text = bytes(base)
        if not text or text[:1] != b"@":                # (the pipeline takes files that begin with '@' only)
            continue
        t = text if text.endswith(b"\n") else text + b"\n"
        mq = (20, 0, 60)[it % 3]
        irr, ds, dq = _frame(E, t, 0, mq)
        try:
            hs, hq = _host(E, tmp_path, text)
            host_ok = True
        except E.EngineError:
            host_ok = False
        if not irr:
            assert host_ok, (it, text[:200])
            assert (ds, dq) == _canon(hs, hq, mq), (it, text[:200])
            regular += 1
        elif host_ok:
            irregular += 1
        else:
            refused += 1
    assert regular > 150 and refused > 100, (regular, irregular, refused)
