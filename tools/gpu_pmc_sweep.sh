#!/bin/bash
# PMC passes (one counter group per run) of tools/sweep.py.  usage: gpu_pmc_sweep.sh TAG KERNEL_SUBSTR [sweep args]
set -u
TAG=${1:-x}; shift || true
KERN=${1:-dense}; shift || true
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
ROOT=$PWD
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_BUSY_avr GRBM_GUI_ACTIVE TA_BUSY_avr" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/sweep.py "$@" > $OUT/g$i.log 2>&1
done
cd - > /dev/null
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:48], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k[0]:50s} {k[1]:34s} mean/dispatch {v / n:14.5g}  n {n}")
PY
