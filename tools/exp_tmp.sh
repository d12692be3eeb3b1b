timeout -k 10 300 python tools/fit_breakdown.py f32 > gpurun_out/fit_breakdown.log 2>&1
timeout -k 10 300 python tools/fit_breakdown.py fp16 >> gpurun_out/fit_breakdown.log 2>&1
cat gpurun_out/fit_breakdown.log
