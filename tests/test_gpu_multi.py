"""The multi-GPU entry (ska.rust_amd/ska_multi.py) against the single-process `ska` executable (`-m gpu`): two ranks share the
one GPU of the test box (gloo carries the exchanges, the engine does the device work), three ranks with uneven shards likewise.
`align` must write the byte-identical file, `distance` the byte-identical table, and the per-rank .skf parts of `build` must merge
into the array the single process builds.  The band form of the distance (skx_planes_distance) is also checked on its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
MULTI = os.path.join(ROOT, "ska.rust_amd", "ska_multi.py")


def _inputs(tmp_path, n=11, length=120_000, seed=9):
    import synth
    anc = synth.ancestor(length, seed=seed)
    files = []
    for i in range(n):
        p = str(tmp_path / f"m{i}.fa")
        synth.to_fasta(synth.sample_stream(anc, i, n, private_snps=60, shared_snps=15, seed=seed), p)
        files.append(p)
    lst = str(tmp_path / "list.txt")
    with open(lst, "w") as f:
        for i, p in enumerate(files):
            f.write(f"m{i}\t{p}\n")
    return files, lst


def _multi(world, port, *args, cwd):
    env = dict(os.environ, SKX_MULTI_BACKEND="gloo", SKX_MULTI_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), MULTI, *args]
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    return r


@pytest.mark.parametrize("world", [2, 3])
def test_multi_align_distance_build_equal_single_process(tmp_path, world):
    files, lst = _inputs(tmp_path)
    wd = str(tmp_path)
    port = 29800 + (os.getpid() % 100) + 10 * world
    # single process
    r = subprocess.run([SKA, "build", "-f", lst, "-o", "one", "-k", "31", "--threads", "4"], cwd=wd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([SKA, "align", "one.skf", "-o", "one.aln"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    for flags, tag in (([], "d0"), (["--min-freq", "0.6"], "d1"), (["--allow-ambiguous"], "d2")):
        assert subprocess.run([SKA, "distance", "one.skf", "-o", f"one.{tag}", *flags], cwd=wd, capture_output=True, timeout=300).returncode == 0
    # several ranks
    _multi(world, port, "align", "-f", lst, "-o", "multi.aln", "-k", "31", "--threads", "2", cwd=wd)
    assert open(os.path.join(wd, "multi.aln"), "rb").read() == open(os.path.join(wd, "one.aln"), "rb").read()
    for flags, tag in (([], "d0"), (["--min-freq", "0.6"], "d1"), (["--allow-ambiguous"], "d2")):
        _multi(world, port + 1, "distance", "-f", lst, "-o", f"multi.{tag}", "-k", "31", "--threads", "2", *flags, cwd=wd)
        assert open(os.path.join(wd, f"multi.{tag}"), "rb").read() == open(os.path.join(wd, f"one.{tag}"), "rb").read(), tag
    _multi(world, port + 2, "build", "-f", lst, "-o", "parts", "-k", "31", "--threads", "2", "--merge", cwd=wd)
    r1 = subprocess.run([SKA, "nk", "parts.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    r2 = subprocess.run([SKA, "nk", "one.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    assert r1.returncode == 0 and r1.stdout == r2.stdout


def test_distance_by_bands_equals_whole(tmp_path):
    import ora
    import skx_engine as E
    E.load_library()
    files, _ = _inputs(tmp_path, n=70, length=30_000, seed=4)          # 70 samples: a 64 x 64 pair tile and a ragged one
    inputs = [(f"m{i}", f, None) for i, f in enumerate(files)]
    for filt in (True, False):
        arr = E.Array.build(inputs, k=31, threads=4)
        constant = arr.filter(0, False, E.FILTER_NO_CONST, False, False, False)
        whole = arr.distance(constant, filt)
        p, wpr, _ = arr.distance_planes(filt)
        S = arr.nsamples
        for bands in (((0, 32), (32, 64), (64, 70)), ((0, 5), (5, 37), (37, 69), (69, 70)), ((0, 1), (1, 70))):     # any band start: tiles begin at i_lo
            got = np.concatenate([E.planes_distance(p, S, wpr, filt, constant, lo, hi) for lo, hi in bands])
            assert np.array_equal(got, whole), bands
        oa = ora.Array.build(inputs, k=31, threads=2)
        oc = oa.filter(0, False, ora.FILTER_NO_CONST, False, False, False)
        od = oa.distance(oc, filt)
        assert oc == constant and np.array_equal(od["match_count"], whole["match_count"]) and np.array_equal(od["mismatch_count"], whole["mismatch_count"])
        assert np.allclose(od["distance"], whole["distance"], rtol=0, atol=1e-6)
