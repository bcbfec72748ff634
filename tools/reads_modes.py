#!/usr/bin/env python
"""tools/reads_modes.py [isolates=128] [gz_isolates=32]: how a batch of BASELINE config 5's read sets (2 x 150 bp at 50x of 5 Mbp, k = 41, --min-count 5,
strict q20) gets to the device -- `ska build` with the readers packing every sample (SKX_KNOBS=reads_raw=1: round 5's form), sending every
sample raw for the device to frame (reads_raw=2), and choosing per sample (the default); every form twice, the .skf files byte-identical; then
the same for .fastq.gz (zlib level 1).  Phases of every run are printed (reader thread-seconds, samples sent raw, GB uploaded)."""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NGZ = int(sys.argv[2]) if len(sys.argv) > 2 else 32
THREADS = os.environ.get("RSC_THREADS", "16")
td = tempfile.mkdtemp(dir="/dev/shm")
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
opts = ["-k", "41", "--min-count", "5", "--min-qual", "20", "--qual-filter", "strict"]


def _write(i, prefix, gz):
    pr = synth.write_read_pair_of(i, N, prefix)
    if gz:
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
            out.append(f + ".gz")
        return pr, out
    return pr, None


def build(tag, lst, knobs):
    ph = os.path.join(td, "ph.json")
    t = time.perf_counter()
    r = subprocess.run([SKA, "build", "-f", lst, "-o", tag, "--threads", THREADS, *opts], cwd=td, capture_output=True, env=dict(os.environ, SKX_KNOBS=knobs, SKX_PHASES=ph))
    dt = time.perf_counter() - t
    assert r.returncode == 0, (tag, r.stderr[-1500:])
    p = json.load(open(ph))
    keep = {k.replace("build.", ""): round(v, 3) for k, v in p.items() if v >= 0.02}
    h = hashlib.sha1(open(os.path.join(td, tag + ".skf"), "rb").read()).hexdigest()[:12]
    os.unlink(os.path.join(td, tag + ".skf"))
    return dt, keep, h


try:
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
        res = list(ex.map(_write, range(N), [os.path.join(td, f"iso{i}") for i in range(N)], [i < NGZ for i in range(N)], chunksize=1))
    print(f"{N} isolates written in {time.perf_counter() - t0:.1f} s ({NGZ} of them also as .gz); cpus {os.cpu_count()}, reader threads {THREADS}", flush=True)
    with open(os.path.join(td, "plain.txt"), "w") as f:
        for i, (pr, _) in enumerate(res):
            f.write(f"iso{i}\t{pr[0]}\t{pr[1]}\n")
    with open(os.path.join(td, "gz.txt"), "w") as f:
        for i, (_, gz) in enumerate(res[:NGZ]):
            f.write(f"iso{i}\t{gz[0]}\t{gz[1]}\n")
    # first: every form on files nobody has read yet (a third of the isolates each: the first read() of freshly written tmpfs pages runs at a
    # third of the rate of the second, and what a user's batch -- or tools/reads_1000.py -- gives the engine is files read for the first time)
    third = N // 3
    modes = (("packed", "reads_raw=1"), ("raw", "reads_raw=2"), ("auto", ""))
    if third >= 2:
        for j, (tag, knobs) in enumerate(modes):
            with open(os.path.join(td, f"fresh{j}.txt"), "w") as f:
                for i in range(j * third, (j + 1) * third):
                    f.write(f"iso{i}\t{res[i][0][0]}\t{res[i][0][1]}\n")
            dt, ph, h = build(tag, f"fresh{j}.txt", knobs)
            print(f"fresh {tag:6s} ({third} isolates, first read of their files): {dt:6.2f} s = {third / dt:6.1f} isolates/s  {ph}", flush=True)
    for lst, n in (("plain.txt", N), ("gz.txt", NGZ)):
        if n < 2:
            continue
        hashes = set()
        for rep in (1, 2):
            for tag, knobs in modes:
                dt, ph, h = build(tag, lst, knobs)
                hashes.add(h)
                print(f"{lst[:-4]:5s} {tag:6s} run {rep}: {dt:6.2f} s = {n / dt:6.1f} isolates/s  {ph}", flush=True)
        print(f"  {lst[:-4]}: the six .skf files are {'IDENTICAL' if len(hashes) == 1 else 'DIFFERENT'}", flush=True)
        assert len(hashes) == 1
finally:
    shutil.rmtree(td, ignore_errors=True)
