"""bench.py's contract that can be checked without a GPU: it refuses to run (loudly) when no gfx950 device is present --
there is no CPU path to fall back to -- and its JSON keys are the ones the driver reads."""
import ast
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a machine without a GPU")
def test_bench_fails_loudly_without_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, timeout=300)
    assert r.returncode != 0
    assert b"needs a gfx950 GPU" in r.stderr + r.stdout and not r.stdout.strip().startswith(b"{")


def test_bench_json_keys_and_oracle_use():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    keys = {k.value for node in ast.walk(tree) if isinstance(node, ast.Dict) for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
    keys |= {n.slice.value for n in ast.walk(tree) if isinstance(n, ast.Subscript) and isinstance(n.slice, ast.Constant) and isinstance(n.slice.value, str)}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "workload", "roofline", "bound", "achieved", "peak", "frac", "traffic", "cpu_baseline", "cores", "kind", "sample"):
        assert k in keys, k
    # the oracle is imported only inside the cpu_baseline leg and the optional --check, never at module level
    top_imports = {a.name for node in tree.body if isinstance(node, (ast.Import, ast.ImportFrom)) for a in node.names}
    assert "ora" not in top_imports
