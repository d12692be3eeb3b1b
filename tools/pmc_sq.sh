#!/bin/bash
# SQ / TA / TCP counters of the legs under one tuning setting.  usage: bash tools/pmc_sq.sh TAG "lean=1"
set -u
TAG=$1; SET=$2; WL=${3:-pl32768}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROG="python3 $PWD/tools/leg_only.py --workload $WL --steps 3 --set $SET"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/a -- $PROG > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- $PROG > $OUT/b.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/c -- $PROG > $OUT/c.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/d -- $PROG > $OUT/d.log 2>&1
cd - > /dev/null
python3 - <<PY
import csv, glob, collections
print("== $TAG: $SET ($WL)")
acc = collections.defaultdict(lambda: [0.0, 0])
for tag in "abcd":
    for f in glob.glob(f"$OUT/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if not any(k in r["Kernel_Name"] for k in ("spmm", "gather3", "dense_tiles")):
                continue
            k = (r["Kernel_Name"][:48], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k in sorted(acc):
    print("  %-50s %-32s %.4e" % (k[0], k[1], acc[k][0] / acc[k][1]))
PY
tail -3 $OUT/a.log
