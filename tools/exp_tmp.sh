timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1; echo "tests rc $?"
tail -2 gpurun_out/r03_gpu_tests.log
timeout -k 10 1000 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc $?"
python tools/bench_summary.py gpurun_out/bench_r03.json 2>/dev/null | head -3
