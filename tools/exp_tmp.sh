rm -f gpurun_out/fused_exp21.log
run() { timeout -k 10 200 python tools/leg_only.py --workload $1 --steps 6 --set fuse=1 >> gpurun_out/fused_exp21.log 2>&1; }
for wl in pl32768d32 pl32768; do
echo "== $wl base" >> gpurun_out/fused_exp21.log; run $wl
echo "== $wl lb3" >> gpurun_out/fused_exp21.log; SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_lb3.so run $wl
echo "== $wl d3" >> gpurun_out/fused_exp21.log; SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_d3.so run $wl
done
grep "wall\|==" gpurun_out/fused_exp21.log | sed 's/exchange1.0.: [0-9.]*, //'
timeout -k 10 300 env SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_d3.so python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
