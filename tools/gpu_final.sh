#!/bin/bash
# Round-end verification on the GPU box: the whole `-m gpu` suite, smoke(), the default bench line.
#   usage (through gpurun, from the repo root):  bash tools/gpu_final.sh TAG      -> gpurun_out/TAG_{gpu_tests,smoke,bench}.log/json
set -u
TAG=${1:-final}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/${TAG}_gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${TAG}_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1 || { tail -20 gpurun_out/${TAG}_smoke.log; exit 1; }
tail -1 gpurun_out/${TAG}_smoke.log
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
python tools/bench_summary.py gpurun_out/${TAG}_bench.json 2>/dev/null | head -60
