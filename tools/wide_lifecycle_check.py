"""The .skf life-cycle with 128-bit keys at BASELINE genome size: two files of `n` samples each (k = 41, 5 Mbp genomes) are
loaded, then merged / weeded / mapped, timed per call; merge(load(A), load(B)) must equal build(A + B).
python tools/wide_lifecycle_check.py [n_per_file]"""
import os, sys, time, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import skx_engine as E
import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = 41
td = tempfile.mkdtemp(prefix="wlc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
anc = synth.ancestor(5_000_000, seed=1)
streams = [synth.sample_stream(anc, i, 2 * n, seed=1).tobytes() for i in range(2 * n)]
names = [f"g{i}" for i in range(2 * n)]


def timed(label, fn):
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print(f"{label}: {dt:.3f} s", flush=True)
    return r


a = E.DictSet.build(streams[:n], K, True).merge(names[:n]); a.save(os.path.join(td, "a.skf"))
b = E.DictSet.build(streams[n:], K, True).merge(names[n:]); b.save(os.path.join(td, "b.skf"))
ab = E.DictSet.build(streams, K, True).merge(names)
wk, wv, wc = ab.export()
print(f"{n} + {n} samples, k = {K}: rows {a.nkmers} + {b.nkmers} -> {ab.nkmers}")
la, lb = E.Array.load(os.path.join(td, "a.skf")), E.Array.load(os.path.join(td, "b.skf"))
m = timed("merge of two loaded arrays", lambda: E.Array.merge([la, lb]))
mk, mv, mc = m.export()
ok = m.names == ab.names and np.array_equal(mk["lo"], wk["lo"]) and np.array_equal(mk["hi"], wk["hi"]) and np.array_equal(mv, wv) and np.array_equal(mc, wc)
print("merge(load(A), load(B)) == build(A + B):", ok)
m2 = timed("merge of two built arrays", lambda: E.Array.merge([a, b]))
k2 = m2.export()[0]
ok2 = np.array_equal(k2["lo"], wk["lo"]) and np.array_equal(k2["hi"], wk["hi"])
ref = os.path.join(td, "anc.fa")
synth.to_fasta(np.concatenate([anc, np.array([10], np.uint8)]), ref)
ks = E.KeySet.from_fasta(ref, K, True)
w1, w2 = E.Array.load(os.path.join(td, "a.skf")), E.Array.load(os.path.join(td, "a.skf"))
r1 = timed("weed of a loaded array", lambda: w1.weed_keys(ks))
r2 = w2.weed_keys(ks, reverse=True)
ok3 = r1 + r2 == la.nkmers and w1.nkmers + w2.nkmers == la.nkmers
aln1 = timed("map of a loaded array (aln)", lambda: la.map(ref))
aln2 = timed("map of the built array (aln)", lambda: a.map(ref))
ok4 = aln1 == aln2
print("checks:", ok, ok2, ok3, ok4)
shutil.rmtree(td)
sys.exit(0 if ok and ok2 and ok3 and ok4 else 1)
