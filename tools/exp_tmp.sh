rm -f gpurun_out/half_exp5.log
timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q > gpurun_out/half_tests10.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/half_tests10.log
run() { timeout -k 10 300 python tools/leg_only.py --steps 6 --storage fp16 "$@" >> gpurun_out/half_exp5.log 2>&1; }
run --workload pl32768d32
run --workload pl65536 --pp
grep "wall" gpurun_out/half_exp5.log
