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
    assert lib.simrank_abi_version() == 7
    rc = lib.simrank_set_tuning(b"no_such_knob", 1)
    assert rc == -1 and b"no_such_knob" in lib.simrank_last_error()
    assert lib.simrank_set_tuning(b"panel", 48) == -1
    assert lib.simrank_set_tuning(b"panel", 0) == 0
    # the diagnostic knobs (wrong results on purpose) are refused in an ordinary process
    import os
    if "SIMRANK_ENABLE_PROBES" not in os.environ:
        assert lib.simrank_set_tuning(b"probe_flags", 1) == -1 and b"SIMRANK_ENABLE_PROBES" in lib.simrank_last_error()
        assert lib.simrank_set_tuning(b"probe_mask", 255) == -1
    assert lib.simrank_set_tuning(b"probe_flags", 0) == 0 and lib.simrank_set_tuning(b"probe_mask", -1) == 0
    v = ctypes.c_int64(-7)
    assert lib.simrank_get_tuning(b"fuse_min", ctypes.byref(v)) == 0 and v.value == 0
    assert lib.simrank_graph_set_dense_terms(None, 3) == -1


def test_host_logic_under_the_sanitizers():
    """make asan: the host side of the library (graph validation, transposed pattern, tiles, dense sets, the
    one-launch plan) compiled for the host only with AddressSanitizer + UBSan and run over random graphs by
    tools/host/host_fuzz.cpp, which also checks every plan entry by entry against the CSR.  No GPU needed."""
    import os
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.LIB_PATH)), "csrc")
    out = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "graphs passed" in out.stdout


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


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """The boundary is a C ABI: the header compiles as C99 with no other include path, and a C
    program linked against the shared library can call it (argument checks only: no device)."""
    import os
    inc = os.path.dirname(_lib.HEADER_PATH)
    src = tmp_path / "use_abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "simrank_hip.h"
int main(void) {
    simrank_graph* g = NULL;
    simrank_epilogue ep;
    memset(&ep, 0, sizeof ep);
    ep.coef = 0.8f;
    if (simrank_abi_version() != SIMRANK_ABI_VERSION) return 1;
    if (simrank_graph_create(0, 4, 0, NULL, NULL, NULL, &g) != SIMRANK_ERR_INVALID) return 2;
    if (simrank_spmm_shard(NULL, NULL, 0, NULL, 0, &ep, 0, 2, NULL, 0, NULL) != SIMRANK_ERR_INVALID) return 3;
    if (simrank_shard_unpack(NULL, 0, NULL, 0, 0, 2, 64, NULL) != SIMRANK_ERR_INVALID) return 4;
    if (!strlen(simrank_last_error())) return 5;
    {   /* the sharded plan's entry points: argument checks only */
        simrank_shardplan* sp = NULL;
        simrank_shardplan_options so;
        memset(&so, 0, sizeof so);
        if (simrank_shardplan_create(4, 0, NULL, NULL, NULL, &so, NULL, NULL, &sp) != SIMRANK_ERR_INVALID) return 6;
        if (simrank_shardplan_step(NULL, 0, 0.0, 1, NULL) != SIMRANK_ERR_INVALID) return 7;
        if (simrank_comm_local_group(0, NULL) != SIMRANK_ERR_INVALID) return 8;
        if (simrank_plan_run_cb(NULL, 1, 0.0, NULL, NULL, NULL, NULL) != SIMRANK_ERR_INVALID) return 9;
        if (simrank_handback_f64(NULL, 0, NULL, 0, 0, 4, NULL, 0, NULL) != SIMRANK_ERR_INVALID) return 10;
        if (simrank_counters_fetch(NULL, NULL, 0, 0, NULL) != SIMRANK_ERR_INVALID) return 11;
    }
    printf("abi %d ok\n", simrank_abi_version());
    return 0;
}
''')
    exe = tmp_path / "use_abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{inc}", str(src),
                         "-o", str(exe), f"-L{libdir}", "-lsimrank_hip", f"-Wl,-rpath,{libdir}"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and "abi 7 ok" in run.stdout, (run.returncode, run.stdout, run.stderr)
