#!/bin/bash
# SQ / TA / TCP / L2 counters of the legs under one tuning setting, one rocprofv3 pass per counter group
# (counters only: no trace domains beside --pmc).  usage: bash tools/pmc_fused.sh TAG "fuse=1" [workload]
set -u
TAG=$1; SET=$2; WL=${3:-pl32768d32}; EXTRA=${4:-}
OUT=$PWD/gpurun_out/pmcf_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROG="python3 $PWD/tools/leg_only.py --workload $WL --steps 2 --set $SET $EXTRA"
cd /tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $PROG > $OUT/g$i.log 2>&1
done
cd - > /dev/null
python3 - <<PY
import csv, glob, collections
print("== $TAG: $SET ($WL)")
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in ("fused", "gather3", "dense_tiles", "half_leg")):
            continue
        k = (r["Kernel_Name"][-52:], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k in sorted(acc):
    print("  %-46s %-32s %.4e" % (k[0], k[1], acc[k][0] / acc[k][1]))
PY
tail -2 $OUT/g1.log
