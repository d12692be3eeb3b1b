#!/bin/bash
# rocprofv3 kernel stats of tools/sweep.py with the given arguments.  usage: gpu_prof_sweep.sh TAG [sweep args]
set -u
TAG=${1:-x}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/sweep.py "$@" > $OUT/stats.log 2>&1
cd - > /dev/null
cat $OUT/stats.log | tail -5
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cut -c1-260 $f | head -12; done
