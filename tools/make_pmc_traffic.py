#!/usr/bin/env python3
"""HBM-side bytes per launch from the rocprofv3 PMC passes of tools/gpu_profile.sh ->
profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of a 16 B/lane
stream (MI355X_MICROARCH.md §HBM; TCC_EA0_RDREQ * 128 B = 2 * FETCH_SIZE * 1024 here as well).
Infinity-Cache hits are included in these counters.

    python3 tools/make_pmc_traffic.py gpurun_out/prof_TAG [workload:ranks]
"""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "pl32768d32:1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: [0.0, 0])
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{out_dir}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])][0] += float(r["Counter_Value"])
            acc[(r["Kernel_Name"], r["Counter_Name"])][1] += 1


def per_launch(match, counter):
    v = [a / n for (k, c), (a, n) in acc.items() if c == counter and match(k)]
    return sum(v)


def traffic(match):
    return (2 * per_launch(match, "FETCH_SIZE") + per_launch(match, "WRITE_SIZE")) * 1024


def is_leg1_gather(k):        # one-launch leg 1 (round 3), or transposed store = template argument MODE 1
    return "fused_trans_kernel" in k or "gather3_kernel<1" in k or "spmm_gather_kernel<4, 8, 1" in k


def is_leg2_gather(k):        # upper-triangle form = MODE 2 (single rank)
    return "gather3_kernel<2" in k or "spmm_gather_kernel<4, 8, 2" in k


def is_dense(k):
    return "dense_tiles_kernel" in k


g1, g2, d = traffic(is_leg1_gather), traffic(is_leg2_gather), traffic(is_dense)
path = os.path.join(root, "profiles", "pmc_traffic.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
rec["_note"] = ("HBM-side bytes per launch from rocprofv3 PMC (tools/gpu_profile.sh + tools/make_pmc_traffic.py): "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 - FETCH_SIZE reads half of a 16 B/lane stream on gfx950 "
                "(MI355X_MICROARCH.md §HBM). Infinity-Cache hits are included. leg1 = fused_trans_kernel (round 3; "
                "rounds 1-2: gather kernel of the remainder + dense_tiles); leg2 = gather kernel, upper-triangle "
                "form, timed exact-count form only (bench.py --exact-only).")
rec[key] = {"leg1": g1 + d, "leg2": g2, "leg1_parts": {"gather": g1, "dense_tiles": d},
            "source": os.path.basename(out_dir.rstrip("/")), "commit": os.environ.get("PMC_COMMIT", "?")}
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec[key], indent=1))
