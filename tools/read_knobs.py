"""reader-thread count vs `ska build` read+upload phase: tools/read_knobs.py"""
import os, subprocess, sys, time, json, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = 1000
td = tempfile.mkdtemp(dir="/dev/shm")
anc = synth.ancestor(5_000_000, seed=1)
files = []
for i in range(n):
    p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); files.append(p)
open(os.path.join(td, "list.txt"), "w").write("".join(f"g{i}\t{p}\n" for i, p in enumerate(files)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
for thr in (16, 32, 32, 64):
    env = dict(os.environ, SKX_PHASES=os.path.join(td, "ph.json"), SKX_DEBUG="1")
    t = time.perf_counter(); r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", str(thr)], cwd=td, capture_output=True, env=env); dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    ph = json.load(open(os.path.join(td, "ph.json")))
    print(thr, "threads: build %.2f s" % dt, {k: round(v, 3) for k, v in ph.items() if k.startswith("build.")}, [l for l in r.stderr.decode().splitlines() if "reader thread" in l])
shutil.rmtree(td)
