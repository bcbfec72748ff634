#!/usr/bin/env python
"""tools/reads_scale_check.py [isolates] [world]: BASELINE config 5's shape through the product -- paired FASTQ isolates (2 x 150 bp at 50x of a
5 Mbp genome, 0.5 % errors, per-cycle Phred profile), k = 41, --min-count 5, strict q20: `ska build` + `ska distance` in one process, and
ska_multi.py distance with `world` ranks sharing the GPU over gloo; the two distance tables must be byte-identical."""
import hashlib, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
td = tempfile.mkdtemp(dir="/dev/shm")
glen, rl, cov = 5_000_000, 150, 50.0
anc = synth.ancestor(glen, seed=1)
comp = np.zeros(256, np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    comp[a] = b
npairs = int(cov * glen / rl / 2)
t0 = time.perf_counter()
with open(os.path.join(td, "list.txt"), "w") as lst:
    for i in range(n):
        rng = np.random.default_rng([1, 99, i])
        g = synth.sample_bases(anc, i, n)
        names = []
        for mate in (0, 1):
            start = rng.integers(0, glen - rl, size=npairs)
            reads = g[start[:, None] + np.arange(rl)[None, :]]
            rev = rng.random(npairs) < 0.5
            reads[rev] = comp[reads[rev][:, ::-1]]
            err = rng.random(reads.shape) < 0.005
            reads[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(err.sum()))]
            prof = np.clip(38 - (np.arange(rl) // 10), 2, 40)
            q = np.clip(prof[None, :] + rng.integers(-6, 3, size=reads.shape), 2, 41).astype(np.uint8) + 33
            q[err] = 33 + 8
            rec = np.empty((npairs, 3 + rl + 3 + rl + 1), np.uint8)        # "@r\n" seq "\n+\n" qual "\n"
            rec[:, 0:3] = np.frombuffer(b"@r\n", np.uint8); rec[:, 3:3 + rl] = reads
            rec[:, 3 + rl:6 + rl] = np.frombuffer(b"\n+\n", np.uint8); rec[:, 6 + rl:6 + 2 * rl] = q; rec[:, -1] = 10
            p = os.path.join(td, f"iso{i}_{mate + 1}.fastq"); rec.tofile(p); names.append(p)
        lst.write(f"iso{i}\t{names[0]}\t{names[1]}\n")
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
for mb in os.environ.get("RSC_BATCH_MB", "").split():
    run(f"ska build with SKX_BUILD_BATCH_MB={mb}", [SKA, "build", "-f", "list.txt", "-o", "three", "--threads", os.environ.get("RSC_THREADS", "16"), *opts], {"SKX_PHASES": os.path.join(td, "ph3.json"), "SKX_BUILD_BATCH_MB": mb, "SKX_DEBUG": "1"})
    print("  phases", open(os.path.join(td, "ph3.json")).read()[:600])
    same = open(os.path.join(td, "one.skf"), "rb").read() == open(os.path.join(td, "three.skf"), "rb").read()
    print("  batched .skf", "IDENTICAL to" if same else "DIFFERENT from", "the one-batch file")
for up in os.environ.get("RSC_UPLOADERS", "").split():
    run(f"ska build with SKX_UPLOADERS={up}", [SKA, "build", "-f", "list.txt", "-o", "two", "--threads", os.environ.get("RSC_THREADS", "16"), *opts], {"SKX_PHASES": os.path.join(td, "ph2.json"), "SKX_UPLOADERS": up})
    print("  phases", open(os.path.join(td, "ph2.json")).read()[:330])
run("ska distance one.skf", [SKA, "distance", "one.skf", "-o", "one.tsv"])
print(f"  = {n / tb:.1f} isolates/s through the executable (files on tmpfs)")
launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", "29642",
          os.path.join(ROOT, "ska.rust_amd", "ska_multi.py")]
run(f"ska_multi distance, {world} ranks on one GPU", launch + ["distance", "-f", "list.txt", "-o", "multi.tsv", "--threads", "8", *opts],
    {"SKX_MULTI_BACKEND": "gloo", "SKX_MULTI_DEVICE": "0"})
h = [hashlib.md5(open(os.path.join(td, f), "rb").read()).hexdigest() for f in ("one.tsv", "multi.tsv")]
print("distances", h, "IDENTICAL" if h[0] == h[1] else "DIFFERENT")
print(open(os.path.join(td, "one.tsv")).read()[:400])
shutil.rmtree(td)
assert h[0] == h[1]
