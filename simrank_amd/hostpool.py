"""Host frames for the dense hand-back (SimRank.py:141, :303: the float64 N x N array a fit returns).

A fresh 8.6 GB array (N = 32768) costs more than its contents: the kernel zeroes every page at the first touch
(2 M page faults, or 4 k huge ones) while the hand-back writes it, and unmapping it when the caller drops the result
takes another 0.4 s — longer than the whole fit (profiles/r04_fit_breakdown_cfg4.log: "free the host copies").  So the
frames are anonymous private mappings of our own that come BACK when the caller's last reference to the array (and to
every view of it: a DataFrame holds one) is gone, and the next result of the same size is written into pages that are
already there — the host twin of the library's device block pool (csrc/api.hip).

At most ``SIMRANK_HOST_POOL_GIB`` (default 9: one config-4 frame) rest here; larger frames, or a second one, are
unmapped as before.  ``trim()`` hands everything back; 0 disables the pool.  Arrays handed out are ordinary writable
float64 ndarrays (``flags.owndata`` is False: the mapping owns the memory and lives as long as any view of it)."""
from __future__ import annotations

import mmap
import os
import threading
import weakref

import numpy as np

_lock = threading.RLock()         # (re-entrant: a frame's finalizer can run inside a collection that starts under the lock)
_free: dict[int, list] = {}        # bytes -> mappings at rest
_rest = 0                          # bytes at rest
MIN_BYTES = 64 << 20               # smaller frames are not worth keeping


def _limit() -> int:
    try:
        return max(0, int(float(os.environ.get("SIMRANK_HOST_POOL_GIB", "9")) * (1 << 30)))
    except ValueError:
        return 9 << 30


def _give_back(mm, nbytes):
    global _rest
    with _lock:
        if _rest + nbytes <= _limit():
            _free.setdefault(nbytes, []).append(mm)
            _rest += nbytes
            return
    try:
        mm.close()
    except (BufferError, ValueError):      # (still exported somewhere: the garbage collector unmaps it later)
        pass


def empty_f64(rows: int, cols: int) -> np.ndarray:
    """Uninitialised float64 [rows, cols], C-contiguous (what ``np.empty`` returns, possibly on recycled pages)."""
    global _rest
    nbytes = int(rows) * int(cols) * 8
    if nbytes < MIN_BYTES or _limit() < nbytes:
        return np.empty((rows, cols), dtype=np.float64)
    mm = None
    with _lock:
        spare = _free.get(nbytes)
        if spare:
            mm = spare.pop()
            _rest -= nbytes
    if mm is None:
        mm = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE)
        try:
            mm.madvise(mmap.MADV_HUGEPAGE)      # (a hint where transparent_hugepage = madvise)
        except (AttributeError, OSError, ValueError):
            pass
    root = np.frombuffer(mm, dtype=np.float64, count=int(rows) * int(cols))
    # every view (the reshaped array below, a DataFrame's block, a slice the caller keeps) holds `root` through .base:
    # when `root` dies nothing can reach the pages any more, and the mapping returns to the pool
    fin = weakref.finalize(root, _give_back, mm, nbytes)
    fin.atexit = False
    return root.reshape(rows, cols)


def trim() -> None:
    """Unmap every frame at rest."""
    global _rest
    with _lock:
        spares = [mm for lst in _free.values() for mm in lst]
        _free.clear()
        _rest = 0
    for mm in spares:
        try:
            mm.close()
        except (BufferError, ValueError):
            pass


def stats() -> tuple[int, int]:
    """(bytes at rest, frames at rest)."""
    with _lock:
        return _rest, sum(len(v) for v in _free.values())
