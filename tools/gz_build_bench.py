#!/usr/bin/env python
"""tools/gz_build_bench.py [isolates=64] [level=1]: `ska build` of paired `.fastq.gz` isolates of BASELINE config 5's shape (2 x 150 bp at 50x of 5 Mbp,
0.55 GB of text an isolate) with the device inflater (default) and with the reader threads' inflater (SKX_KNOBS=reads_gz=1), alternating, the
.skf files compared byte for byte; prints isolates/s and the phases of each run."""
import json, os, subprocess, sys, tempfile, time, zlib
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
LEVEL = int(sys.argv[2]) if len(sys.argv) > 2 else 1
td = tempfile.mkdtemp(dir="/dev/shm")
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")


def _write(i):
    out = []
    for f in synth.write_read_pair_of(i, 1000, os.path.join(td, f"iso{i}")):
        c = zlib.compressobj(LEVEL, zlib.DEFLATED, 31)
        with open(f, "rb") as src, open(f + ".gz", "wb") as dst:
            while True:
                b = src.read(8 << 20)
                if not b:
                    break
                dst.write(c.compress(b))
            dst.write(c.flush())
        os.unlink(f)
        out.append(f + ".gz")
    return out


t = time.perf_counter()
with ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
    pairs = list(ex.map(_write, range(N), chunksize=1))
gb = sum(os.path.getsize(f) for p in pairs for f in p) / 1e9
print(f"{N} isolates as .fastq.gz (level {LEVEL}): {gb:.1f} GB, written in {time.perf_counter() - t:.0f} s", flush=True)
with open(os.path.join(td, "list.txt"), "w") as lst:
    for i, (a, b) in enumerate(pairs):
        lst.write(f"iso{i}\t{a}\t{b}\n")
runs = os.environ.get("GZB_RUNS", "device;host;device;host").split(";")
ref = None
for r, tag in enumerate(runs):
    env = dict(os.environ)
    knobs = [k for k in env.get("SKX_KNOBS", "").split(",") if k]
    if tag.startswith("host"):
        knobs.append("reads_gz=1")
    knobs += [k for k in tag.split(":")[1:] if k]
    env["SKX_KNOBS"] = ",".join(knobs)
    ph = os.path.join(td, f"ph{r}.json")
    env["SKX_PHASES"] = ph
    t = time.perf_counter()
    p = subprocess.run([SKA, "build", "-f", "list.txt", "-o", f"out{r}", "-k", "41", "--min-count", "5", "--min-qual", "20", "--qual-filter", "strict", "--threads", os.environ.get("GZB_THREADS", "16")],
                       cwd=td, capture_output=True, env=env)
    dt = time.perf_counter() - t
    assert p.returncode == 0, p.stderr[-1500:]
    phases = json.load(open(ph))
    skf = open(os.path.join(td, f"out{r}.skf"), "rb").read()
    same = ref is None or skf == ref
    ref = ref or skf
    print(f"run {r} [{tag}] knobs={env['SKX_KNOBS']!r}: {dt:.2f} s = {N / dt:.1f} isolates/s, .skf {len(skf)} bytes, equal to the first: {same}", flush=True)
    print("   ", {k.replace("build.", ""): round(v, 3) for k, v in phases.items() if v >= 0.05 or "samples" in k}, flush=True)
    assert same
import shutil
shutil.rmtree(td, ignore_errors=True)
