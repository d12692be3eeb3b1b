#!/usr/bin/env python3
"""Soak of whole fits on a real GPU: the body of tests/test_gpu_parity.py::test_randomized_against_oracle
(random size / density / class / sharding / mode, every result against the float64 oracle at 1e-5) over many
seeds.  Not part of the test suite: `python3 tools/soak_fits.py [first_seed] [count]`, log in profiles/."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import test_randomized_against_oracle as one_case    # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0, ok, refused = time.time(), 0, 0
# SOAK_KNOBS="fuse_sym=1,fuse_steps=1": library knobs for the whole run (e.g. the one-launch leg 2 on every graph that has a set)
if os.environ.get("SOAK_KNOBS"):
    from simrank_amd.engine import HipOps as _H
    _H(0).set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["SOAK_KNOBS"].split(","))})
    print("knobs:", os.environ["SOAK_KNOBS"], flush=True)
for seed in range(first, first + count):
    try:
        one_case(seed)
        ok += 1
    except ValueError as e:
        # (a random prior whose size does not match the node count of a graph with isolated ids: the estimator
        # refuses it as the reference does)
        if "broadcast" not in str(e):
            raise
        refused += 1
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak_fits: seeds {first}..{first + count - 1}: {ok} fits matched the oracle, {refused} refused like the reference")
