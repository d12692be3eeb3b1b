#!/bin/bash
# rocprofv3 kernel stats + PMC passes for the bench workload.  usage: bash tools/gpu_profile.sh TAG [bench args]
set -u
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --exact-only $*"
HALF="python3 $PWD/tools/leg_only.py --workload pl65536 --pp --storage fp16 --steps 5"
cd /tmp
if [ -z "${ONLY_HALF:-}" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $BENCH > $OUT/pmc_l2.log 2>&1
rocprofv3 --pmc TCC_BUSY_avr GRBM_GUI_ACTIVE TA_BUSY_avr --output-format csv -d $OUT/pmc_busy -- $BENCH > $OUT/pmc_busy.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/pmc_ea -- $BENCH > $OUT/pmc_ea.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- $BENCH > $OUT/pmc_tcp.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
fi
# config 5 on fp16-held matrices (half.hip): kernel stats + HBM-side bytes of the two legs
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/half_stats -- $HALF > $OUT/half_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/half_fetch -- $HALF > $OUT/half_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/half_write -- $HALF > $OUT/half_write.log 2>&1
rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/half_busy -- $HALF > $OUT/half_busy.log 2>&1
cd - > /dev/null
find $OUT -name "*.csv" | head -20
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2500])
for f in glob.glob("$OUT/half_stats/**/*kernel_stats.csv", recursive=True):
    print("== config 5, fp16-held matrices (tools/leg_only.py --workload pl65536 --pp --storage fp16)")
    print(open(f).read()[:1500])
for tag in ("pmc_l2", "pmc_busy", "pmc_ea", "pmc_tcp", "pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write",
            "half_fetch", "half_write", "half_busy"):
    for f in glob.glob(f"$OUT/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if not any(k in r["Kernel_Name"] for k in ("spmm", "gather3", "gemm", "dense_tiles", "fused", "half_leg")):
                continue
            k = (r["Kernel_Name"][:62], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(tag, k, "mean per dispatch", v / n, "n", n)
PY
python3 $PWD/tools/make_pmc_traffic.py $OUT ${PMC_KEY:-pl32768d32:1} || true
