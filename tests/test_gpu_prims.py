"""The engine's own device primitives (csrc/skx_prims.hip: radix sorts, scans, selections, unique, the segmented OR-scan) against the host's
std::stable_sort / loops on seeded data, at sizes around every tile boundary (a wave's 1 024 keys, a block's 4 096 items, several passes of
the carry scan).  They replaced rocPRIM in round 5 and have no entry of their own in the C ABI: the library exports one self-check."""
import ctypes as C

import pytest

import skx_engine as E

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 70_001, 1_048_577, 17_000_003])
def test_primitives_against_the_host(n):
    lib = E.load_library()
    f = lib.skx_debug_prims_selftest
    f.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
    f.restype = C.c_int
    assert f(0, n, 1234 + n) == 0
