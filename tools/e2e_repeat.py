#!/usr/bin/env python
"""tools/e2e_repeat.py [n] [threads]: ska build / ska align x.skf / ska align *.fa on n synthetic 5 Mbp assemblies (FASTA on tmpfs), each several
times in a row, wall clock and the engine's phase table of every run: how much of a run is the box's state after the run before"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
thr = sys.argv[2] if len(sys.argv) > 2 else "64"
td = tempfile.mkdtemp(dir="/dev/shm")
anc = synth.ancestor(5_000_000, seed=1)
files = []
for i in range(n):
    p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); files.append(p)
open(os.path.join(td, "list.txt"), "w").write("".join(f"g{i}\t{p}\n" for i, p in enumerate(files)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
def run(tag, args):
    ph = os.path.join(td, "ph.json")
    t = time.perf_counter(); r = subprocess.run([SKA, *args], cwd=td, capture_output=True, env=dict(os.environ, SKX_PHASES=ph)); dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    p = json.load(open(ph))
    if os.environ.get("SKX_DEBUG"):
        print("".join("    " + l + "\n" for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("[skx]")), end="")
    print(f"{tag} {dt:.3f} s  " + " ".join(f"{k.split('.')[-1]}={v:.3f}" for k, v in p.items() if v >= 0.02), flush=True)
for rep in range(3):
    run("build ", ["build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", thr])
for rep in range(3):
    run("align ", ["align", "all.skf", "-o", "aln.fa", "--threads", thr])
for rep in range(2):
    run("single", ["align", "-o", "aln2.fa", "--threads", thr, *files])
run("build ", ["build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", thr])
run("align ", ["align", "all.skf", "-o", "aln.fa", "--threads", thr])
shutil.rmtree(td)
