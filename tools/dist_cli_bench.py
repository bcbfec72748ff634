"""`ska distance` at BASELINE size through the executable: tools/dist_cli_bench.py [n_genomes]  (build first, then distance x.skf)"""
import os, subprocess, sys, time, json, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
td = tempfile.mkdtemp(dir="/dev/shm")
anc = synth.ancestor(5_000_000, seed=1)
files = []
for i in range(n):
    p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); files.append(p)
open(os.path.join(td, "list.txt"), "w").write("".join(f"g{i}\t{p}\n" for i, p in enumerate(files)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
def run(args):
    env = dict(os.environ, SKX_PHASES=os.path.join(td, "ph.json"))
    t = time.perf_counter(); r = subprocess.run([SKA, *args], cwd=td, capture_output=True, env=env); dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    return dt, json.load(open(os.path.join(td, "ph.json")))
run(["build", "-f", "list.txt", "-o", "all", "-k", os.environ.get("DCB_K", "31"), "--threads", "32"])
for flags in ([], ["--allow-ambiguous"], ["--min-freq", "0.9"]):
    dt, ph = run(["distance", "all.skf", "-o", "d.tsv", *flags])
    print(n, "samples, ska distance", " ".join(flags), ": %.2f s wall," % dt, "%d pairs," % (n * (n - 1) // 2), "table %.1f MB" % (os.path.getsize(os.path.join(td, "d.tsv")) / 1e6), {k: round(v, 3) for k, v in ph.items()})
shutil.rmtree(td)
