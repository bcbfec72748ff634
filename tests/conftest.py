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
