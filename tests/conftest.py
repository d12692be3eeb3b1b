import json
import os
import sys

import numpy as np
import pandas as pd
import pytest

import tests.pydriver  # noqa: F401,E402  (installs the Python choreography — the tests' double — as estimators.PYTHON_SOLVER)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible() -> bool:
    try:
        from simrank_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A bare ``pytest`` on a host without a HIP device skips the gpu tests instead of failing
    them; with ``-m gpu`` asked for explicitly they run (and fail loudly: no CPU fallback)."""
    if "gpu" in (config.getoption("-m") or "") or _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (gpu tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def free_port() -> int:
    """A TCP port nobody listens on, for a torch.distributed rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)


class Golden:
    """One committed fixture: inputs and the reference's outputs."""

    def __init__(self, name):
        self.name = name
        self.meta = MANIFEST[name]
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.frame = pd.DataFrame({c: z["in_" + c] for c in self.meta["columns"]})
        self.args = [z[f"arg{i}"] for i in range(self.meta["nargs"])]
        self.out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
        self.kwargs = dict(self.meta["kwargs"])
        self.cls = self.meta["cls"]
        self.raises = self.meta.get("raises")
        self.k = self.meta.get("k")
        self.stdout = self.meta.get("stdout")


def golden_names(*classes):
    return sorted(n for n, m in MANIFEST.items() if not classes or m["cls"] in classes)


@pytest.fixture
def golden():
    return Golden
