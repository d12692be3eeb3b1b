rm -f gpurun_out/fused_exp16.log
for w in pl32768d32 pl32768 er8192 er32768; do
timeout -k 10 160 python tools/leg_only.py --workload $w --steps 10 --set fuse=1 >> gpurun_out/fused_exp16.log 2>&1
done
grep "^fuse" gpurun_out/fused_exp16.log
