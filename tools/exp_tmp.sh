rm -f gpurun_out/half_exp4.log
timeout -k 10 900 python -m pytest tests/test_gpu_half.py -x -q > gpurun_out/half_tests6.log 2>&1; echo "tests rc $?"
tail -4 gpurun_out/half_tests6.log
run() { timeout -k 10 300 python tools/leg_only.py --steps 6 --storage fp16 "$@" >> gpurun_out/half_exp4.log 2>&1; }
for wl in "--workload pl32768d32" "--workload pl65536 --pp"; do
run $wl
run $wl --set fuse_min=4
run $wl --set fuse_min=6
done
grep "wall" gpurun_out/half_exp4.log
