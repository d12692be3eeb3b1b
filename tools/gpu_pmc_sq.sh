#!/bin/bash
# SQ-level PMC passes (instruction mix, waits, LDS conflicts) for one tools/sweep.py configuration.
# usage: bash tools/gpu_pmc_sq.sh TAG [sweep args]
set -u
TAG=${1:-sq}; shift || true
OUT=$PWD/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/sweep.py $*"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/d -- $CMD > $OUT/d.log 2>&1
cd - > /dev/null
python3 - <<PY
import csv, glob, collections
for tag in "abcd":
    for f in glob.glob(f"$OUT/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "spmm" not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"][14:60], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(k[0], k[1], f"{v / n:.4g}", "n", n)
PY
