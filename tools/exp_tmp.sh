timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
rm -f gpurun_out/fused_exp27.log
run() { timeout -k 10 300 python tools/leg_only.py --workload $1 --steps 8 $2 $3 $4 >> gpurun_out/fused_exp27.log 2>&1; }
run pl32768d32; run pl32768; run er8192; run pl65536 --pp; run pl65536 --pp --storage fp16; run pl32768d32 --storage fp16
grep "wall" gpurun_out/fused_exp27.log | sed 's/exchange1.0.: [0-9.]*, //'
