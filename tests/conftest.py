import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "ska.rust_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _knob_string(cur, name, value=None):
    items = [x for x in (cur or "").split(",") if x and x.split("=")[0] != name]
    if value is not None:
        items.append(f"{name}={value}")
    return ",".join(items)


def set_knob(monkeypatch, name, value):
    """one test / measurement knob of the engine (SKX_KNOBS=name=value,...: the library's only switch of this kind)"""
    import os
    monkeypatch.setenv("SKX_KNOBS", _knob_string(os.environ.get("SKX_KNOBS"), name, value))


def del_knob(monkeypatch, name):
    import os
    monkeypatch.setenv("SKX_KNOBS", _knob_string(os.environ.get("SKX_KNOBS"), name))


def pytest_collection_modifyitems(config, items):
    """GPU runs: torch's copy of the HIP runtime must be the first one loaded in the process (the copy loaded second finds no GPU), whichever
    test files were selected and in whichever order -- so it is initialised here, before any test loads the engine."""
    if "not gpu" in (config.getoption("markexpr") or "") or not any(it.get_closest_marker("gpu") for it in items):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
