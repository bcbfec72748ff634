"""CPU-side checks of the drop-in boundary: libskx.so loads, exports every symbol the headers declare,
and fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import pytest

import skx_engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("skx.h", "skx_host.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(sk[xh]_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    declared = _declared()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by libskx.so"
    assert declared == set(E.SYMBOLS), declared ^ set(E.SYMBOLS)


def test_version_string():
    assert E.load_library().skx_version() == b"0.5.2"


def test_sample_name_rule():
    # io_utils.rs:31-46
    assert E.sample_name("/a/b/test_1.fa") == "test_1"
    assert E.sample_name("x.fastq.gz") == "x"
    assert E.sample_name("dir/x.FASTA") == "x"
    assert E.sample_name("dir/x.fa.gz") == "dir/x.fa.gz"
    assert E.sample_name("x.fasta.fa") == "x.fasta"


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_device_fails_loudly():
    with pytest.raises(E.EngineError) as ei:
        E.Context(0)
    assert ei.value.code == E.ENODEV


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "ska.rust_amd")
    for dp, _, files in os.walk(pkg):
        if os.path.basename(dp) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "ska_oracle" not in txt and "libska_oracle" not in txt and "import ora" not in txt, os.path.join(dp, f)


def test_skf_peek_k_reads_the_first_fields_only():
    """skx_skf_peek_k (include/skx.h): a file's k without loading it -- what lets `ska merge` ask for the right key width at once (lib.rs:635-661
    tries u64, then u128).  No device is touched."""
    import ctypes as C
    import glob
    import skx_engine as eng
    lib = eng.load_library()
    lib.skx_skf_peek_k.argtypes = [C.c_char_p]
    lib.skx_skf_peek_k.restype = C.c_int
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    found = {os.path.basename(f): lib.skx_skf_peek_k(f.encode()) for f in glob.glob(os.path.join(root, "**", "*.skf"), recursive=True)}
    assert found.get("merge_k41.skf") == 41 and found.get("merge_k9.skf") == 9 and found.get("merge.skf") == 17, found
    assert lib.skx_skf_peek_k(b"/nonexistent/file.skf") == 0
    assert lib.skx_skf_peek_k(__file__.encode()) == 0                      # not an .skf at all
