#!/usr/bin/env python
"""tools/reads_scale_check.py [isolates] [world]: BASELINE config 5's shape through the product -- paired FASTQ isolates (2 x 150 bp at 50x of a
5 Mbp genome, 0.5 % errors, per-cycle Phred profile), k = 41, --min-count 5, strict q20: `ska build` + `ska distance` in one process, and
`ska distance --gpus <world>` with the ranks sharing the GPU (world 0: skipped); the two distance tables must be byte-identical.  (Parity
against the oracle at this shape: tests/test_gpu_zz_full_size.py::test_config5_flow_eight_isolates_at_size.)"""
import hashlib, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
td = tempfile.mkdtemp(dir="/dev/shm")
glen, rl, cov = 5_000_000, 150, 50.0
npairs = int(cov * glen / rl / 2)
t0 = time.perf_counter()
# the isolates are simulated by worker processes (synth.write_read_pair: ~10 s of numpy each; this process never opens the GPU, so forks are fine)
from concurrent.futures import ProcessPoolExecutor
workers = int(os.environ.get("RSC_WORKERS", str(min(n, 64, os.cpu_count() or 1))))
def _write(i, n_, prefix):
    pr = synth.write_read_pair_of(i, n_, prefix)
    if os.environ.get("RSC_GZ"):                                  # the read sets as .fastq.gz (zlib level 1)
        import zlib
        out = []
        for f in pr:
            c = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(f, "rb") as src, open(f + ".gz", "wb") as dst:
                while True:
                    b = src.read(8 << 20)
                    if not b:
                        break
                    dst.write(c.compress(b))
                dst.write(c.flush())
            os.unlink(f)
            out.append(f + ".gz")
        pr = out
    return pr
with ProcessPoolExecutor(max_workers=workers) as ex:
    pairs = list(ex.map(_write, range(n), [n] * n, [os.path.join(td, f"iso{i}") for i in range(n)], chunksize=1))
with open(os.path.join(td, "list.txt"), "w") as lst:
    for i, (f1, f2) in enumerate(pairs):
        lst.write(f"iso{i}\t{f1}\t{f2}\n")
sys.stdout.flush()
print(f"{n} isolates written in {time.perf_counter() - t0:.1f} s ({npairs} pairs each)", flush=True)
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
def run(tag, cmd, env=None):
    t = time.perf_counter(); r = subprocess.run(cmd, cwd=td, capture_output=True, env=dict(os.environ, **(env or {}))); dt = time.perf_counter() - t
    assert r.returncode == 0, (tag, r.stderr[-1500:])
    print(f"{tag}: {dt:.2f} s", flush=True)
    if os.environ.get("RSC_DEBUG"): print(r.stderr.decode()[-3000:])
    return dt
opts = ["-k", "41", "--min-count", "5", "--min-qual", "20", "--qual-filter", "strict"]
tb = run("ska build (one process)", [SKA, "build", "-f", "list.txt", "-o", "one", "--threads", os.environ.get("RSC_THREADS", "16"), *opts], {"SKX_PHASES": os.path.join(td, "ph.json"), "SKX_DEBUG": "1"} if os.environ.get("RSC_DEBUG") else {"SKX_PHASES": os.path.join(td, "ph.json")})
print("  phases", open(os.path.join(td, "ph.json")).read())
if os.environ.get("RSC_SPOT"):                                  # isolates of this very run against the oracle's dictionaries of their files (CPU, one thread each)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ora
    from concurrent.futures import ThreadPoolExecutor
    m = min(int(os.environ["RSC_SPOT"]), n)
    idx = sorted({(n - 1) * j // max(1, m - 1) for j in range(m)})
    t = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(idx)) as pool:
        dicts = list(pool.map(lambda i: ora.Dict.from_files(41, pairs[i][0], pairs[i][1], True, ora.qual(5, 20, ora.QUAL_STRICT)).export(), idx))
    arr = ora.Array.load(os.path.join(td, "one.skf"))
    keys, var, _ = arr.export()
    order = np.lexsort((keys["lo"], keys["hi"]))
    for i, (ok, ob) in zip(idx, dicts):
        col = var[order, i]
        have = col != ord("-")
        same = int(have.sum()) == len(ok) and np.array_equal(keys["lo"][order][have], ok["lo"]) and np.array_equal(keys["hi"][order][have], ok["hi"]) and np.array_equal(col[have], ob)
        print(f"  isolate {i}: {len(ok)} split k-mers in the oracle's dictionary; column {i} of one.skf {'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
        assert same
    print(f"  ({len(idx)} isolates of the {n} spot-checked against the oracle in {time.perf_counter() - t:.1f} s)", flush=True)
    del arr, keys, var, dicts
for mb in os.environ.get("RSC_BATCH_MB", "").split():
    run(f"ska build with SKX_BUILD_BATCH_MB={mb}", [SKA, "build", "-f", "list.txt", "-o", "three", "--threads", os.environ.get("RSC_THREADS", "16"), *opts], {"SKX_PHASES": os.path.join(td, "ph3.json"), "SKX_BUILD_BATCH_MB": mb, "SKX_DEBUG": "1"})
    print("  phases", open(os.path.join(td, "ph3.json")).read()[:600])
    same = open(os.path.join(td, "one.skf"), "rb").read() == open(os.path.join(td, "three.skf"), "rb").read()
    print("  batched .skf", "IDENTICAL to" if same else "DIFFERENT from", "the one-batch file")
if os.environ.get("RSC_ONE_SHOT"):                               # the same build without the reader / kernel pipeline: the .skf must be the same bytes
    run("ska build, one-shot form (SKX_KNOBS=no_reads_pipeline)", [SKA, "build", "-f", "list.txt", "-o", "oneshot", "--threads", os.environ.get("RSC_THREADS", "16"), *opts],
        {"SKX_PHASES": os.path.join(td, "ph5.json"), "SKX_KNOBS": "no_reads_pipeline"})
    print("  phases", open(os.path.join(td, "ph5.json")).read()[:420])
    same = open(os.path.join(td, "one.skf"), "rb").read() == open(os.path.join(td, "oneshot.skf"), "rb").read()
    print("  one-shot .skf", "IDENTICAL to" if same else "DIFFERENT from", "the pipelined build's")
    assert same
for th in os.environ.get("RSC_THREADS_SWEEP", "").split():      # the same build with other reader-thread counts
    run(f"ska build --threads {th}", [SKA, "build", "-f", "list.txt", "-o", "sweep", "--threads", th, *opts], {"SKX_PHASES": os.path.join(td, "ph4.json")})
    print("  phases", open(os.path.join(td, "ph4.json")).read()[:420])
run("ska distance one.skf", [SKA, "distance", "one.skf", "-o", "one.tsv"])
print(f"  = {n / tb:.1f} isolates/s through the executable (files on tmpfs)")
print(f"  one.skf {os.path.getsize(os.path.join(td, 'one.skf')) / 1e9:.2f} GB, one.tsv {os.path.getsize(os.path.join(td, 'one.tsv')) / 1e6:.1f} MB")
if world > 0:
    # the same job as `world` ranks sharing the GPU (ska --gpus N, exchanges through skx_comm_* on the host-staged transport)
    run(f"ska distance --gpus {world} (ranks sharing one GPU)", [SKA, "distance", "--gpus", str(world), "-f", "list.txt", "-o", "multi.tsv", "--threads", "8", *opts],
        {"SKX_COMM": "local", "SKX_DEVICE": "0"})
    h = [hashlib.md5(open(os.path.join(td, f), "rb").read()).hexdigest() for f in ("one.tsv", "multi.tsv")]
    print("distances", h, "IDENTICAL" if h[0] == h[1] else "DIFFERENT")
    assert h[0] == h[1]
print(open(os.path.join(td, "one.tsv")).read()[:400])
shutil.rmtree(td)
