import json
import os
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
