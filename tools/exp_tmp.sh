rm -f gpurun_out/fused_exp22.log
run() { timeout -k 10 200 python tools/leg_only.py --workload $1 --steps 8 --set fuse=1 >> gpurun_out/fused_exp22.log 2>&1; }
for wl in pl32768d32 pl32768; do
echo "== $wl base" >> gpurun_out/fused_exp22.log; run $wl
echo "== $wl prio2" >> gpurun_out/fused_exp22.log; SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_prio2.so run $wl
echo "== $wl prio3" >> gpurun_out/fused_exp22.log; SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_prio3.so run $wl
done
grep "wall\|==" gpurun_out/fused_exp22.log | sed 's/exchange1.0.: [0-9.]*, //'
