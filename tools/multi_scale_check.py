#!/usr/bin/env python
"""tools/multi_scale_check.py [n] [world]: the product launcher at size on ONE GPU -- `world` ranks share it over gloo (what the driver's
multi-GPU nodes do over RCCL): ska_multi.py align / distance on n synthetic 5 Mbp assemblies must give byte for byte what one `ska align` /
`ska distance` process gives (the alignment's columns are the same set in the same order: both derive the global rows in engine order)."""
import hashlib, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
td = tempfile.mkdtemp(dir="/dev/shm")
anc = synth.ancestor(5_000_000, seed=1)
files = []
with open(os.path.join(td, "list.txt"), "w") as f:
    for i in range(n):
        p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); files.append(p); f.write(f"g{i}\t{p}\n")
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
def run(tag, cmd, env=None):
    t = time.perf_counter(); r = subprocess.run(cmd, cwd=td, capture_output=True, env=dict(os.environ, **(env or {}))); dt = time.perf_counter() - t
    assert r.returncode == 0, (tag, r.stderr[-1500:])
    print(f"{tag}: {dt:.2f} s", flush=True)
def md5(p):
    h = hashlib.md5()
    with open(os.path.join(td, p), "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""): h.update(b)
    return h.hexdigest(), os.path.getsize(os.path.join(td, p))
launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", "29641",
          os.path.join(ROOT, "ska.rust_amd", "ska_multi.py")]
genv = {"SKX_MULTI_BACKEND": "gloo", "SKX_MULTI_DEVICE": "0"}
run("ska build (one process)", [SKA, "build", "-f", "list.txt", "-o", "one", "-k", "31", "--threads", "32"])
run("ska align one.skf", [SKA, "align", "one.skf", "-o", "one.aln", "--threads", "32"])
run("ska distance one.skf", [SKA, "distance", "one.skf", "-o", "one.tsv", "--threads", "32"])
run(f"ska_multi align, {world} ranks on one GPU", launch + ["align", "-f", "list.txt", "-o", "multi.aln", "--threads", "16", "--report", "rep_align.json"], genv)
run(f"ska_multi distance, {world} ranks on one GPU", launch + ["distance", "-f", "list.txt", "-o", "multi.tsv", "--threads", "16", "--report", "rep_dist.json"], genv)
a1, a2, d1, d2 = md5("one.aln"), md5("multi.aln"), md5("one.tsv"), md5("multi.tsv")
print("alignment", a1, a2, "IDENTICAL" if a1 == a2 else "DIFFERENT")
print("distances", d1, d2, "IDENTICAL" if d1 == d2 else "DIFFERENT")
for r in ("rep_align.json", "rep_dist.json"):
    print(r, open(os.path.join(td, r)).read())
shutil.rmtree(td)
assert a1 == a2 and d1 == d2
