timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "restricted or config5" 2>&1 | tail -3
timeout -k 10 300 python tools/fit_breakdown.py f32 2>&1 | tail -9
