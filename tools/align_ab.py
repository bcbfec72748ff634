#!/usr/bin/env python
"""tools/align_ab.py [n] "<ENV=..>" "<ENV=..>" ...: `ska align x.skf` (and `ska distance x.skf`) on n synthetic 5 Mbp assemblies under different
environments, alternating, four rounds; wall clock + phase table of every run"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
envs = sys.argv[2:] or ["X=0"]
td = tempfile.mkdtemp(dir="/dev/shm")
anc = synth.ancestor(5_000_000, seed=1)
with open(os.path.join(td, "list.txt"), "w") as f:
    for i in range(n):
        p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); f.write(f"g{i}\t{p}\n")
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
def run(tag, args, env):
    ph = os.path.join(td, "ph.json")
    e = dict(os.environ, SKX_PHASES=ph); e.update(dict(x.split("=", 1) for x in env.split()))
    t = time.perf_counter(); r = subprocess.run([SKA, *args], cwd=td, capture_output=True, env=e); dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    p = json.load(open(ph))
    print(f"{tag:9s} {env:22s} {dt:.3f} s  " + " ".join(f"{k.split('.')[-1]}={v:.3f}" for k, v in p.items() if v >= 0.004), flush=True)
run("build", ["build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", "64"], "X=0")
time.sleep(3)
for rep in range(4):
    for env in envs:
        if os.path.exists(os.path.join(td, "aln.fa")): os.unlink(os.path.join(td, "aln.fa")); time.sleep(float(os.environ.get("AB_SLEEP", "1.0")))      # (truncating 4.9 GB on open costs 0.4 s)
        run("align", ["align", "all.skf", "-o", "aln.fa", "--threads", "64"], env)
for rep in range(2):
    for env in envs:
        run("distance", ["distance", "all.skf", "-o", "d.tsv", "--threads", "64"], env)
shutil.rmtree(td)
