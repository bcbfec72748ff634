"""write_fasta knobs on one loaded array: tools/fasta_knobs.py (builds 1000 genomes once, then times `ska align x.skf` variants)"""
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
def run(args, extra_env):
    env = dict(os.environ, SKX_PHASES=os.path.join(td, "ph.json"), **extra_env)
    t = time.perf_counter(); r = subprocess.run([SKA, *args], cwd=td, capture_output=True, env=env); dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    return dt, json.load(open(os.path.join(td, "ph.json")))
run(["build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", "64"], {})
for env in ({}, {"SKX_FASTA_KT": "8"}, {"SKX_FASTA_KT": "16"}, {"SKX_FASTA_KT": "2"}, {"SKX_FASTA_PREALLOC": "1"}, {"SKX_FASTA_PREALLOC": "1", "SKX_FASTA_KT": "8"}, {"SKX_FASTA_NB": "6", "SKX_FASTA_KT": "8"}, {"SKX_NO_MMAP_OUTPUT": "1"}):
    if os.path.exists(os.path.join(td, "aln.fa")): os.unlink(os.path.join(td, "aln.fa"))
    dt, ph = run(["align", "all.skf", "-o", "aln.fa"], env)
    print(env, "align %.2f s" % dt, {k: round(v, 3) for k, v in ph.items() if k.startswith("fasta") or k == "align.write_fasta"})
shutil.rmtree(td)
