#!/usr/bin/env python
"""tools/gz_device_bench.py [level=1]: one read file of BASELINE config 5's shape (150 bp reads at 50x of 5 Mbp: 275 MB of text) as .fastq.gz through
the device inflater's test hook (skx_debug_gz_inflate), checked against zlib; prints the device time of the decode kernels (find, decode, maps,
groups) and of the text + CRC kernels for a few chunk sizes.  Under rocprofv3 --kernel-trace --stats the per-kernel split."""
import os, sys, time, zlib, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
import skx_engine as eng
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
td = "/dev/shm/gzdb"
os.makedirs(td, exist_ok=True)
a, b = synth.write_read_pair_of(0, 1000, os.path.join(td, "r"))
text = open(a, "rb").read()
t = time.perf_counter()
c = zlib.compressobj(level, zlib.DEFLATED, 31)
blob = c.compress(text) + c.flush()
print(f"text {len(text) / 1e6:.1f} MB, gz level {level}: {len(blob) / 1e6:.1f} MB ({time.perf_counter() - t:.1f} s to deflate)", flush=True)
t = time.perf_counter(); ref = zlib.decompress(blob, 31); print(f"zlib inflate on one core: {time.perf_counter() - t:.2f} s", flush=True)
lib = eng.load_library()
ctx = eng.default_context()
lib.skx_debug_gz_inflate.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
lib.skx_debug_gz_inflate.restype = C.c_int
out = C.create_string_buffer(len(text) + 64)
for knobs in (os.environ.get("GZB_KNOBS", "").split(";") if os.environ.get("GZB_KNOBS") else ["gz_chunk_kb=64", "gz_chunk_kb=32", "gz_chunk_kb=128", "gz_chunk_kb=256", "gz_chunk_kb=64,gz_group=8", "gz_chunk_kb=64,gz_group=128"]):
    os.environ["SKX_KNOBS"] = knobs
    total, status, members = C.c_uint64(), C.c_uint32(), C.c_uint32()
    ms = (C.c_double * 2)()
    t = time.perf_counter()
    rc = lib.skx_debug_gz_inflate(ctx.h, blob, len(blob), len(text), out, len(text) + 64, C.byref(total), C.byref(status), C.byref(members), ms)
    dt = time.perf_counter() - t
    ok = rc == 0 and status.value == 0 and out.raw[:total.value] == text
    print(f"{knobs}: rc {rc} status {status.value} members {members.value} equal {ok}; decode {ms[0]:.2f} ms, text+crc {ms[1]:.2f} ms (call {dt:.2f} s)", flush=True)
