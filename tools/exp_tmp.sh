rm -f gpurun_out/half_exp2.log
timeout -k 10 600 python -m pytest tests/test_gpu_half.py -x -q > gpurun_out/half_tests2.log 2>&1; echo "tests rc $?"
tail -8 gpurun_out/half_tests2.log
timeout -k 10 200 python tools/leg_only.py --workload pl32768d32 --steps 10 --storage fp16 >> gpurun_out/half_exp2.log 2>&1
timeout -k 10 300 python tools/leg_only.py --workload pl65536 --pp --steps 5 --storage fp16 >> gpurun_out/half_exp2.log 2>&1
grep "wall" gpurun_out/half_exp2.log
