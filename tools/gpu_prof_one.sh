#!/bin/bash
# rocprofv3 kernel stats of ONE program.  usage: bash tools/gpu_prof_one.sh TAG python3 <script> [args]   (csv under gpurun_out/prof_TAG)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$*"
ROOT=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
cd $ROOT
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cut -c1-160 "$f" | sed -n 1,14p; done
