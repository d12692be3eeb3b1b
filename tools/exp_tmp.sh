timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "topk" 2>&1 | tail -5
timeout -k 10 300 python tools/fit_breakdown.py f32 2>&1 | tail -8
