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
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[-14:]:
        print(f'{r["Kernel_Name"][:60]:60s} start {(int(r["Start_Timestamp"])-t0)/1e6:10.3f} ms  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:8.3f} ms  grid {r.get("Grid_Size_X", r.get("Grid_Size",""))}')
PY
