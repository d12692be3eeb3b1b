#!/bin/bash
# rocprofv3 kernel stats + HBM-side counters (FETCH_SIZE, WRITE_SIZE; separate --pmc passes, counters only) for BASELINE
# configs 2, 3, 4, 5 (f32 and fp16-held) at the current commit.   usage: bash tools/gpu_profile_configs.sh TAG
# -> gpurun_out/prof_TAG/<config>/{stats,fetch,write}/...csv; tools/make_config_profiles.py turns them into
#    profiles/<TAG>_rocprof_summary.md, profiles/<TAG>_kernel_stats_<config>.csv and profiles/pmc_traffic.json
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_$TAG
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LEG="python3 $ROOT/tools/leg_only.py --exact"
run_config() {          # name, program...
  local name=$1; shift
  mkdir -p $OUT/$name
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/stats -- "$@" > $OUT/$name/stats.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/fetch -- "$@" > $OUT/$name/fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/write -- "$@" > $OUT/$name/write.log 2>&1
  cd $ROOT
  echo "profiled $name" >> $OUT/progress.log
}
run_config cfg2_er8192 $LEG --workload er8192 --steps 20
run_config cfg3_ml1m $LEG --workload ml1m --steps 10
run_config cfg4_pl32768d32 python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --exact-only
run_config cfg5_pl65536_pp $LEG --workload pl65536 --pp --steps 4
run_config cfg5_pl65536_pp_fp16 $LEG --workload pl65536 --pp --storage fp16 --steps 4
# L2 / busy counters of the headline configuration (as rounds 2-4 reported them)
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --exact-only"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/cfg4_pl32768d32/l2 -- $BENCH > $OUT/cfg4_pl32768d32/l2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/cfg4_pl32768d32/sq -- $BENCH > $OUT/cfg4_pl32768d32/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/cfg4_pl32768d32/tcp -- $BENCH > $OUT/cfg4_pl32768d32/tcp.log 2>&1
cd $ROOT
python3 tools/make_config_profiles.py $OUT $TAG
