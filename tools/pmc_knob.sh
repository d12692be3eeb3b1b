#!/bin/bash
# L2 hit/miss + fabric reads of the legs under one tuning setting.
# usage: bash tools/pmc_knob.sh TAG "balance=1" [workload]
set -u
TAG=$1; SET=$2; WL=${3:-pl32768}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROG="python3 $PWD/tools/leg_only.py --workload $WL --steps 3 --set $SET"
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- $PROG > $OUT/l2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum --output-format csv -d $OUT/ea -- $PROG > $OUT/ea.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PROG > $OUT/stats.log 2>&1
cd - > /dev/null
python3 - <<PY
import csv, glob, collections
print("== $TAG: $SET ($WL)")
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("spmm", "gather3", "dense_tiles")):
            print("  %-70s calls %s avg %.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
acc = collections.defaultdict(lambda: [0.0, 0])
for tag in ("l2", "ea"):
    for f in glob.glob(f"$OUT/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if not any(k in r["Kernel_Name"] for k in ("spmm", "gather3", "dense_tiles")):
                continue
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
names = sorted({k[0] for k in acc})
for nme in names:
    g = lambda c: acc[(nme, c)][0] / max(1, acc[(nme, c)][1])
    h, m, ea = g("TCC_HIT_sum"), g("TCC_MISS_sum"), g("TCC_EA0_RDREQ_sum")
    print("  %-60s L2 req %.3e hit rate %.3f fabric reads %.2f GB" % (nme, h + m, h / max(1.0, h + m), ea * 128 / 1e9))
PY
