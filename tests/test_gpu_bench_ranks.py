"""bench.py's N > 1 path (`-m gpu`): two ranks share the one GPU of the test box over gloo (SKX_BENCH_BACKEND / SKX_BENCH_DEVICE), the
launch line is the driver's.  The JSON line must carry the whole-job rate, the weak-scaling label and the exchange sizes, and the
sharded row set must be the one a single rank derives for the same samples (rows_U)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env=None, launcher=()):
    cmd = [sys.executable, *launcher, os.path.join(ROOT, "bench.py"), "--genomes", "24", "--genome-len", "300000", "--steps", "2", "--warmup", "1",
           "--cpu-genomes", "0", "--no-e2e", *([] if "--pmc-genomes" in extra else ["--no-pmc"]), *extra]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_two_ranks_on_one_gpu():
    port = str(29600 + os.getpid() % 300)
    two = _bench(["--gpus", "2"], env={"SKX_BENCH_BACKEND": "gloo", "SKX_BENCH_DEVICE": "0"},      # gloo rendezvous -> the engine's host-staged transport
                 launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port))
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["steps"] == 2 and two["value"] > 0
    assert two["config"]["samples_total"] == 48 if "samples_total" in two["config"] else True
    assert two.get("exchange_per_step_rank0"), two.keys()
    one = _bench(["--gpus", "1", "--genomes", "48"])
    assert one["n_gpus"] == 1
    assert two["config"]["rows_U"] == one["config"]["rows_U"], (two["config"], one["config"])      # the sharded row set is the single-rank row set


def test_bench_sharded_path_on_rccl_at_world_one():
    """SKX_BENCH_FORCE_SHARDED=1 on the default backend: bench.py's N > 1 step (skx_keyset_allgather + skx_array_reduce_stats over an
    RCCL communicator) runs on the one-GPU box before an 8-GPU node ever sees it, and derives the rows of the unsharded step"""
    forced = _bench(["--gpus", "1"], env={"SKX_BENCH_FORCE_SHARDED": "1", "MASTER_PORT": str(29400 + os.getpid() % 90)})
    plain = _bench(["--gpus", "1"])
    assert forced["exchange_per_step_rank0"]["transport"] == "rccl"
    assert forced["config"]["rows_U"] == plain["config"]["rows_U"] and forced["config"]["rows_kept"] == plain["config"]["rows_kept"]


def test_bench_gpus_without_a_launcher_starts_its_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it must start two ranks itself (never one rank printing n_gpus: 1)"""
    env = {"SKX_BENCH_BACKEND": "gloo", "SKX_BENCH_DEVICE": "0", "MASTER_PORT": str(29650 + os.getpid() % 200)}
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genomes", "16", "--genome-len", "200000", "--steps", "1", "--warmup", "1",
           "--cpu-genomes", "0", "--no-e2e"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(base, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0


def test_bench_refuses_a_world_that_is_not_gpus():
    """a launcher that started one rank for --gpus 2 gets a non-zero exit, not a line"""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genomes", "8", "--genome-len", "100000", "--steps", "1", "--warmup", "0",
           "--cpu-genomes", "0", "--no-e2e"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300, env=dict(base, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "refusing" in r.stderr


def test_bench_measures_the_extraction_kernels_traffic_in_the_run():
    """roofline.traffic comes from rocprofv3 --pmc passes of this very run (FETCH_SIZE x 2 + WRITE_SIZE of the extraction kernel's launches over a
    child process), not from a file under profiles/: at least the W + 1 bytes a base leaves in the regions, and no more than a few times that"""
    line = _bench(["--gpus", "1", "--pmc-genomes", "12", "--no-check", "--no-distance"])
    rl = line["roofline"]
    assert rl["traffic_source"].startswith("measured in this run"), rl["traffic_source"]
    assert 0.3 * rl["algorithmic_bytes_per_launch"] < rl["traffic"] < 6.0 * rl["algorithmic_bytes_per_launch"], rl
