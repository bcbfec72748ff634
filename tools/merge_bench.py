#!/usr/bin/env python
"""tools/merge_bench.py [n=1000] [batches=4] [k=41]: `ska merge` of `batches` .skf files built from n synthetic 5 Mbp assemblies (n / batches each), with the
files loaded side by side (the default) and one after the other (SKX_KNOBS=serial_loads), twice each, round robin: wall clock and phases; the merged
files must be byte-identical."""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = sys.argv[3] if len(sys.argv) > 3 else "41"
td = tempfile.mkdtemp(dir="/dev/shm")
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
try:
    anc = synth.ancestor(5_000_000, seed=1)
    for b in range(nb):
        with open(os.path.join(td, f"list{b}.txt"), "w") as f:
            for i in range(n * b // nb, n * (b + 1) // nb):
                p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); f.write(f"g{i}\t{p}\n")
        r = subprocess.run([SKA, "build", "-f", f"list{b}.txt", "-o", f"batch{b}", "-k", k, "--threads", "32"], cwd=td, capture_output=True)
        assert r.returncode == 0, r.stderr[-400:]
    print("batch files:", [round(os.path.getsize(os.path.join(td, f"batch{b}.skf")) / 1e9, 2) for b in range(nb)], "GB", flush=True)
    hashes = set()
    for rep in range(2):
        for knobs in ("", "serial_loads=1"):
            ph = os.path.join(td, "ph.json")
            t = time.perf_counter()
            r = subprocess.run([SKA, "merge", *[f"batch{b}.skf" for b in range(nb)], "-o", "all"], cwd=td, capture_output=True, env=dict(os.environ, SKX_KNOBS=knobs, SKX_PHASES=ph))
            dt = time.perf_counter() - t
            assert r.returncode == 0, r.stderr[-400:]
            hashes.add(hashlib.sha1(open(os.path.join(td, "all.skf"), "rb").read()).hexdigest())
            os.unlink(os.path.join(td, "all.skf"))
            print(f"[{knobs:14s}] ska merge {dt:.2f} s ", {a: round(v, 2) for a, v in json.load(open(ph)).items() if v >= 0.05}, flush=True)
    print("merged files", "IDENTICAL" if len(hashes) == 1 else "DIFFERENT")
    assert len(hashes) == 1
finally:
    shutil.rmtree(td, ignore_errors=True)
