rm -f gpurun_out/fused_exp28.log
run() { timeout -k 10 300 python tools/leg_only.py --workload $1 --steps 8 $2 >> gpurun_out/fused_exp28.log 2>&1; }
for v in base ge4096 ge8192 ge12288 mr128 mr512 mu32 mu96 base; do
  if [ $v = base ]; then unset SIMRANK_LIB; else export SIMRANK_LIB=$PWD/gpurun_variants/libsimrank_hip_$v.so; fi
  for wl in pl32768d32 pl32768 er32768; do echo "== $v $wl" >> gpurun_out/fused_exp28.log; run $wl; done
  echo "== $v pl65536pp" >> gpurun_out/fused_exp28.log; run pl65536 --pp
done
grep "wall\|==" gpurun_out/fused_exp28.log | sed 's/exchange1.0.: [0-9.]*, //' | paste - - | awk '{print $2, $3, $6, $7}' | tr -d "{',"
