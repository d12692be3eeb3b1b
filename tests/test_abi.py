"""The C-ABI shared library on a machine without a GPU: it loads, exports exactly what
include/simrank_hip.h declares, and every device call fails loudly (no CPU fallback)."""
import ctypes
import re
import subprocess

import pytest

from simrank_amd import _lib


def _declared():
    text = open(_lib.HEADER_PATH).read()
    return sorted(set(re.findall(r"^SIMRANK_API [\w \*]+?\b(simrank_\w+)\(", text, flags=re.M)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.PROTOTYPES)
    assert len(_declared()) >= 29


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True).stdout
    exported = sorted(set(re.findall(r" T (simrank_\w+)", out)))
    assert exported == _declared()          # nothing else leaks out (-fvisibility=hidden)


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.simrank_abi_version() == 2
    rc = lib.simrank_set_tuning(b"no_such_knob", 1)
    assert rc == -1 and b"no_such_knob" in lib.simrank_last_error()
    assert lib.simrank_set_tuning(b"panel", 48) == -1
    assert lib.simrank_set_tuning(b"panel", 0) == 0


def test_argument_checks_need_no_device():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.simrank_graph_create(0, 4, 0, None, None, None, ctypes.byref(h)) == -1
    assert lib.simrank_spmm(None, None, 0, 0, None, 0, 0, 0, 0, None, None) == -1
    assert lib.simrank_gemm_nt(4, 4, 4, None, 4, None, 4, None, 4, None, None) == -1


def test_product_path_fails_loudly_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    import pandas as pd
    import simrank_amd.SimRank as SRA
    from simrank_amd._lib import SimRankHipError
    with pytest.raises(SimRankHipError, match="no CPU fallback"):
        SRA.SimRank().fit(pd.DataFrame({"from": [1, 2], "to": [2, 1]}), verbose=False)
