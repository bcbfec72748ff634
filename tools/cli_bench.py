#!/usr/bin/env python
"""End-to-end CLI timing ("build+align as two CLI invocations through a .skf on tmpfs and as the single
`ska align *.fa` form"): tools/cli_bench.py [n_genomes] [genome_len] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
threads = sys.argv[3] if len(sys.argv) > 3 else "32"
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
anc = synth.ancestor(glen, seed=1)
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    files = []
    for i in range(n):
        p = os.path.join(td, f"g{i}.fa")
        synth.to_fasta(synth.sample_stream(anc, i, n), p)
        files.append(p)
    with open(os.path.join(td, "list.txt"), "w") as f:
        for i, p in enumerate(files):
            f.write(f"g{i}\t{p}\n")

    def run(*args):
        t0 = time.perf_counter()
        r = subprocess.run([SKA, *args], cwd=td, capture_output=True)
        assert r.returncode == 0, r.stderr[-500:]
        if os.environ.get("SKX_DEBUG"):
            sys.stderr.write("".join(l + "\n" for l in r.stderr.decode().splitlines() if l.startswith("[skx]")))
        return time.perf_counter() - t0, r.stdout

    tb, _ = run("build", "-f", "list.txt", "-o", "all", "-k", "31", "--threads", threads)
    size = os.path.getsize(os.path.join(td, "all.skf"))
    ta, _ = run("align", "all.skf", "-o", "aln.fa")             # to a file on tmpfs: a pipe into this script would be the bottleneck
    aln = open(os.path.join(td, "aln.fa"), "rb").read() if n <= 200 else b"x" * os.path.getsize(os.path.join(td, "aln.fa"))
    ts, _ = run("align", "--threads", threads, "-o", "aln2.fa", *files) if n <= 200 else (float("nan"), None)
    print(f"{n} genomes x {glen} bp, {threads} reader threads: ska build {tb:.2f} s (.skf {size / 1e6:.1f} MB), ska align {ta:.2f} s "
          f"({len(aln) / 1e6:.1f} MB alignment), single `ska align *.fa` {ts:.2f} s; {n / (tb + ta):.1f} genomes/s end to end through the CLI")
