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


def _ska_ranks(world, *args, cwd, env=None):
    """the executable's own launcher: `ska <cmd> --gpus N ...` starts N ranks of itself; SKX_COMM=local + SKX_DEVICE=0 let them share the GPU"""
    e = dict(os.environ, SKX_COMM="local", SKX_DEVICE="0", **(env or {}))
    r = subprocess.run([SKA, args[0], "--gpus", str(world), *args[1:]], cwd=cwd, capture_output=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    return r


@pytest.mark.parametrize("world", [2, 3])
def test_ska_gpus_flag_equals_single_process(tmp_path, world):
    """`ska build|align|distance --gpus N` (ranks forked by the executable, exchanges through skx_comm_*) == the single process"""
    files, lst = _inputs(tmp_path, n=10, length=90_000, seed=21)
    wd = str(tmp_path)
    assert subprocess.run([SKA, "build", "-f", lst, "-o", "one", "--threads", "4"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([SKA, "align", "one.skf", "-o", "one.aln", "--filter", "no-filter", "-m", "0.5"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([SKA, "distance", "one.skf", "-o", "one.dist"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    _ska_ranks(world, "align", "-f", lst, "-o", "multi.aln", "--filter", "no-filter", "-m", "0.5", "--threads", "2", cwd=wd)
    assert open(os.path.join(wd, "multi.aln"), "rb").read() == open(os.path.join(wd, "one.aln"), "rb").read()
    _ska_ranks(world, "distance", "-f", lst, "-o", "multi.dist", "--threads", "2", cwd=wd)
    assert open(os.path.join(wd, "multi.dist"), "rb").read() == open(os.path.join(wd, "one.dist"), "rb").read()
    r = _ska_ranks(world, "distance", *files, "--threads", "2", cwd=wd)                 # sequence files given directly, table on rank 0's stdout
    names_from_files = open(os.path.join(wd, "one.dist"), "rb").read()
    assert r.stdout.count(b"\n") == names_from_files.count(b"\n")
    _ska_ranks(world, "build", "-f", lst, "-o", "parts", "--threads", "2", "--merge", cwd=wd)
    r1 = subprocess.run([SKA, "nk", "parts.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    r2 = subprocess.run([SKA, "nk", "one.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    assert r1.returncode == 0 and r1.stdout == r2.stdout
    _ska_ranks(world, "build", "-f", lst, "-o", "keep", "--threads", "2", cwd=wd)          # without --merge: one valid .skf per rank
    parts = [f"keep.part{r}of{world}.skf" for r in range(world)]
    assert all(os.path.exists(os.path.join(wd, p)) for p in parts)
    assert subprocess.run([SKA, "merge", "-o", "joined", *parts], cwd=wd, capture_output=True, timeout=300).returncode == 0
    r3 = subprocess.run([SKA, "nk", "joined.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    assert r3.stdout == r2.stdout
    # a rank that fails takes the job down with the engine's message instead of leaving its peers in a collective
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write(open(lst).read().replace(files[-1], files[-1] + ".missing"))
    r = subprocess.run([SKA, "align", "--gpus", str(world), "-f", bad, "-o", "x.aln"], cwd=wd, capture_output=True, timeout=300,
                       env=dict(os.environ, SKX_COMM="local", SKX_DEVICE="0", SKX_COMM_TIMEOUT_S="30"))
    assert r.returncode != 0


def test_rccl_branches_run_at_world_one(tmp_path):
    """The RCCL transport (ncclCommInitRank, ncclAllGather, ncclAllReduce on the engine's stream) executes on a one-GPU box at world
    size 1: through the executable (`--gpus 1`), through ska_multi.py on the nccl backend, and through the ABI directly."""
    files, lst = _inputs(tmp_path, n=9, length=80_000, seed=31)
    wd = str(tmp_path)
    assert subprocess.run([SKA, "build", "-f", lst, "-o", "one", "--threads", "4"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([SKA, "align", "one.skf", "-o", "one.aln"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([SKA, "distance", "one.skf", "-o", "one.dist", "--allow-ambiguous"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    env = {k: v for k, v in os.environ.items() if not k.startswith("SKX_COMM")}
    r = subprocess.run([SKA, "align", "--gpus", "1", "-f", lst, "-o", "g1.aln", "--threads", "2"], cwd=wd, capture_output=True, env=dict(env, SKX_DEBUG="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    assert open(os.path.join(wd, "g1.aln"), "rb").read() == open(os.path.join(wd, "one.aln"), "rb").read()
    r = subprocess.run([SKA, "distance", "--gpus", "1", "-f", lst, "-o", "g1.dist", "--allow-ambiguous"], cwd=wd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    assert open(os.path.join(wd, "g1.dist"), "rb").read() == open(os.path.join(wd, "one.dist"), "rb").read()
    # ska_multi.py, default backend (nccl == RCCL), one rank
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
           str(29900 + os.getpid() % 90), MULTI, "distance", "-f", lst, "-o", "m1.dist", "--allow-ambiguous", "--threads", "2", "--report", "rep.json"]
    r = subprocess.run(cmd, cwd=wd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    assert open(os.path.join(wd, "m1.dist"), "rb").read() == open(os.path.join(wd, "one.dist"), "rb").read()
    import json
    assert json.load(open(os.path.join(wd, "rep.json")))["transport"] == "rccl"
    # the ABI directly: an RCCL communicator of one rank, in a process of its own so that its exit (RCCL / HIP teardown) is checked too
    r = subprocess.run([sys.executable, "-c", _ABI_RCCL_WORLD_ONE, lst], cwd=wd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0 and b"rccl world-1 ok" in r.stdout, (r.returncode, r.stderr[-1500:].decode(errors="replace"))
    assert b"double free" not in r.stderr and b"corruption" not in r.stderr, r.stderr[-1500:]


_ABI_RCCL_WORLD_ONE = r"""
import os, sys
import numpy as np
ROOT = %r
sys.path[:0] = [os.path.join(ROOT, "ska.rust_amd"), os.path.join(ROOT, "tests")]
import torch             # before the engine opens RCCL: a process holds one copy of librccl, and PyTorch brings its own
import skx_engine as E
E.load_library()
names, files = zip(*[l.split()[:2] for l in open(sys.argv[1])])
names, files = list(names), list(files)
ctx = E.Context(0)
comm = E.Comm.rccl(0, 1, E.comm_unique_id(), ctx=ctx)
assert (comm.rank, comm.world) == (0, 1)
ds = E.DictSet.from_files([(f, None) for f in files], 31, True, threads=4, ctx=ctx)
ks = ds.union_keys(notes=True)
rows = comm.keyset_allgather(ks)
assert len(rows) == len(ks)
arr = ds.assemble(rows, names)
want = E.Array.build([(n, f, None) for n, f in zip(names, files)], k=31, threads=4, ctx=ctx)
comm.reduce_stats(arr, len(files))
assert all(np.array_equal(x, y) for x, y in zip(arr.export(), want.export()))
constant = arr.filter(0, False, E.FILTER_NO_CONST, False, False, False)
want.filter(0, False, E.FILTER_NO_CONST, False, False, False)
for filt in (True, False):
    assert np.array_equal(comm.distance_sharded(arr, len(files), filt, constant), want.distance(constant, filt))
t = torch.arange(1000, dtype=torch.int32, device="cuda:0")
out = torch.empty_like(t)
comm.allgather_device(t.data_ptr(), out.data_ptr(), t.numel() * 4)
comm.allreduce_u32_device(t.data_ptr(), t.numel())
ctx.sync()
assert torch.equal(out, t) and int(t[999]) == 999
comm.free()
print("rccl world-1 ok")
""" % ROOT


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


def test_selftest_two_ranks_and_one(tmp_path):
    """`ska selftest --gpus N`: the pre-flight of a sharded job (id hand-off or directory, all-reduce, gather, all-gather of unequal key tables)"""
    ska = os.path.join(ROOT, "ska.rust_amd", "ska")
    env = dict(os.environ, SKX_COMM="local", SKX_DEVICE="0")
    r = subprocess.run([ska, "selftest", "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "2 rank(s)" in r.stderr and ": ok" in r.stderr, r.stderr[-1500:]
    # BASELINE config 4's exchange at its size (round 6): ~8 M keys per rank, unequal, the union's row count known in advance
    shape = [l for l in r.stderr.splitlines() if "config-4 shape" in l]
    assert shape and shape[-1].endswith(": ok") and "union of 2 tables" in shape[-1], r.stderr[-1500:]
    r = subprocess.run([ska, "selftest"], capture_output=True, text=True, timeout=120, env=dict(os.environ, SKX_DEVICE="0"))
    assert r.returncode == 0 and "1 rank(s)" in r.stderr and "config-4 shape" in r.stderr, r.stderr[-1500:]
    # a stray SKX_WORLD does not turn `ska nk` into a sharded job
    r = subprocess.run([ska, "nk", os.path.join(ROOT, "tests", "golden", "input", "merge.skf")], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, SKX_WORLD="2", SKX_RANK="0", SKX_DEVICE="0"))
    assert r.returncode == 0, r.stderr[-1500:]


def test_ranks_with_unrelated_samples_equal_single_process(tmp_path):
    """Three ranks whose samples share nothing: the rows of a rank's hash range are three times its own (its pieces are laid over all ranks'
    rows in several passes of the rows kernel), most cells of its columns are '-'.  align and distance == the single process, byte for byte."""
    import synth
    files = []
    for i in range(6):
        anc = synth.ancestor(150_000, seed=100 + i // 2)               # pairs of related samples; the pairs are unrelated
        p = str(tmp_path / f"u{i}.fa")
        synth.to_fasta(synth.sample_stream(anc, i % 2, 2, private_snps=40, shared_snps=5, seed=100 + i // 2), p)
        files.append(p)
    lst = str(tmp_path / "list.txt")
    with open(lst, "w") as f:
        for i, p in enumerate(files):
            f.write(f"u{i}\t{p}\n")
    wd = str(tmp_path)
    for cmd, out, extra in (("align", "aln", ["--min-freq", "0.3", "--filter", "no-filter"]), ("distance", "dist", ["--min-freq", "0"])):
        r = subprocess.run([SKA, cmd, "-f", lst, "-o", f"one.{out}", "-k", "31", *extra], cwd=wd, capture_output=True, timeout=300)
        if r.returncode != 0:                                            # (align / distance of sequence files in one process go through build first)
            assert subprocess.run([SKA, "build", "-f", lst, "-o", "one", "-k", "31"], cwd=wd, capture_output=True, timeout=300).returncode == 0
            r = subprocess.run([SKA, cmd, "one.skf", "-o", f"one.{out}", *extra], cwd=wd, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        _ska_ranks(3, cmd, "-f", lst, "-o", f"multi.{out}", "-k", "31", *extra, cwd=wd)
        assert open(os.path.join(wd, f"multi.{out}"), "rb").read() == open(os.path.join(wd, f"one.{out}"), "rb").read(), cmd


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs: the RCCL transport at world 2 (skipped on the one-GPU test box, runs wherever two devices exist)")
def test_rccl_world_two_equals_single_process(tmp_path):
    """First contact with a second device should not be a scaling run: `ska selftest --gpus 2`, then skx_keyset_allgather (ncclAllGather of
    padded key tables from two devices), skx_array_reduce_stats (ncclAllReduce + ncclAllGather) and skx_array_distance_sharded (plane
    all-gather, grouped ncclSend / ncclRecv to rank 0) through `ska align | distance | build --gpus 2` over RCCL -- byte-identical to the
    single process.  Stands where merge_ska_dict.rs:354-417 joins its worker threads."""
    files, lst = _inputs(tmp_path, n=13, length=150_000, seed=41)
    wd = str(tmp_path)
    env = {k: v for k, v in os.environ.items() if not k.startswith("SKX_COMM") and k != "SKX_DEVICE"}
    r = subprocess.run([SKA, "selftest", "--gpus", "2"], cwd=wd, capture_output=True, env=env, timeout=300)
    assert r.returncode == 0 and b": ok" in r.stderr, r.stderr[-1500:].decode(errors="replace")
    assert subprocess.run([SKA, "build", "-f", lst, "-o", "one", "--threads", "4"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([SKA, "align", "one.skf", "-o", "one.aln"], cwd=wd, capture_output=True, timeout=300).returncode == 0
    for flags, tag in (([], "d0"), (["--allow-ambiguous"], "d2")):
        assert subprocess.run([SKA, "distance", "one.skf", "-o", f"one.{tag}", *flags], cwd=wd, capture_output=True, timeout=300).returncode == 0
    r = subprocess.run([SKA, "align", "--gpus", "2", "-f", lst, "-o", "two.aln", "--threads", "2"], cwd=wd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    assert open(os.path.join(wd, "two.aln"), "rb").read() == open(os.path.join(wd, "one.aln"), "rb").read()
    for flags, tag in (([], "d0"), (["--allow-ambiguous"], "d2")):
        r = subprocess.run([SKA, "distance", "--gpus", "2", "-f", lst, "-o", f"two.{tag}", *flags], cwd=wd, capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
        assert open(os.path.join(wd, f"two.{tag}"), "rb").read() == open(os.path.join(wd, f"one.{tag}"), "rb").read(), tag
    r = subprocess.run([SKA, "build", "--gpus", "2", "-f", lst, "-o", "parts", "--threads", "2", "--merge"], cwd=wd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:].decode(errors="replace")
    r1 = subprocess.run([SKA, "nk", "parts.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    r2 = subprocess.run([SKA, "nk", "one.skf", "--full-info"], cwd=wd, capture_output=True, timeout=300)
    assert r1.returncode == 0 and r1.stdout == r2.stdout


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs")
def test_bench_two_gpus_over_rccl_reports_its_ranks(tmp_path):
    """bench.py --gpus 2 as the driver launches it: the selftest pre-flight ran, the line reports the RCCL communicator's own rank count and a
    device per rank, and the sharded row set is the single-rank one"""
    import json
    bench = os.path.join(ROOT, "bench.py")
    common = ["--genomes", "24", "--genome-len", "300000", "--steps", "2", "--warmup", "1", "--cpu-genomes", "0", "--no-e2e", "--no-pmc"]
    env = {k: v for k, v in os.environ.items() if not k.startswith("SKX_") and k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(29700 + os.getpid() % 200), bench, "--gpus", "2", *common]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    two = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["selftest"] and ": ok" in two["selftest"]["report"]
    assert sorted(d["hip_device"] for d in two["rank_devices"]) == [0, 1]
    r = subprocess.run([sys.executable, bench, "--gpus", "1", *common[:1], "48", *common[2:]], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    one = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert two["config"]["rows_U"] == one["config"]["rows_U"]
