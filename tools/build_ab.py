#!/usr/bin/env python
"""tools/build_ab.py [n=1000] "<SKX_KNOBS a>" "<SKX_KNOBS b>" ...: `ska build` (+ `ska align x.skf`) on n synthetic 5 Mbp assemblies under different
run-time knobs, round robin, three rounds on one box: wall clock and the phases above 20 ms of every run."""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
variants = sys.argv[2:] or ["", "no_prewarm=1"]
td = tempfile.mkdtemp(dir="/dev/shm")
try:
    anc = synth.ancestor(5_000_000, seed=1)
    files = []
    for i in range(n):
        p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); files.append(p)
    open(os.path.join(td, "list.txt"), "w").write("".join(f"g{i}\t{p}\n" for i, p in enumerate(files)))
    SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
    def run(tag, knobs, args):
        ph = os.path.join(td, "ph.json")
        t = time.perf_counter(); r = subprocess.run([SKA, *args], cwd=td, capture_output=True, env=dict(os.environ, SKX_PHASES=ph, SKX_KNOBS=knobs)); dt = time.perf_counter() - t
        assert r.returncode == 0, r.stderr[-300:]
        p = json.load(open(ph))
        print(f"[{knobs:14s}] {tag} {dt:.3f} s  " + " ".join(f"{k.split('.')[-1]}={v:.3f}" for k, v in p.items() if v >= 0.02), flush=True)
    for rep in range(3):
        for v in variants:
            run("build", v, ["build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", "64"])
            run("align", v, ["align", "all.skf", "-o", "aln.fa", "--threads", "64"])
finally:
    shutil.rmtree(td, ignore_errors=True)
