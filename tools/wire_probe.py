#!/usr/bin/env python3
"""Cost of the fp16 wire format per rank and update: simrank_narrow_h16 + simrank_widen_h16 over buffers of the size
one rank exchanges (N x N / P floats for exchange 1; + 50 % for the mirrored tiles of the half form), by HIP events."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simrank_amd._lib import check                              # noqa: E402
from simrank_amd.engine import HipOps                           # noqa: E402

ops = HipOps(0)
lib, st = ops.lib, ops.stream
scale = C.c_float(16384.0)
for n_nodes, P in ((32768, 8), (32768, 4), (65536, 8)):
    n = n_nodes * n_nodes // P
    src, dst = ops._malloc(4 * n), ops._malloc(2 * n)
    check(lib.simrank_memset(C.c_void_p(src), 0, 4 * n, st), "memset")
    e = [ops.event() for _ in range(3)]
    for rep in range(3):
        ops.record(e[0])
        check(lib.simrank_narrow_h16(C.c_void_p(src), C.c_void_p(dst), n, scale, st), "narrow")
        ops.record(e[1])
        check(lib.simrank_widen_h16(C.c_void_p(dst), C.c_void_p(src), n, scale, st), "widen")
        ops.record(e[2])
        ops.event_synchronize(e[2])
    a, b = ops.elapsed_ms(e[0], e[1]), ops.elapsed_ms(e[1], e[2])
    print(f"N={n_nodes} P={P}: exchange 1 of {4 * n / 2**20:.0f} MiB per rank -> narrow {a:.3f} ms "
          f"({6e-6 * n / a:.0f} GB/s), widen {b:.3f} ms ({6e-6 * n / b:.0f} GB/s); "
          f"link bytes per rank {4 * n * (P - 1) / P / 2**20:.0f} -> {2 * n * (P - 1) / P / 2**20:.0f} MiB", flush=True)
    for ev in e:
        ops.event_destroy(ev)
    ops._free(src)
    ops._free(dst)
