rm -f gpurun_out/fused_exp15.log
for w in pl32768d32 er8192 er32768 pl65536; do
for s in fuse=1; do
timeout -k 10 160 python tools/leg_only.py --workload $w --steps 10 --set $s >> gpurun_out/fused_exp15.log 2>&1
done; done
grep "^fuse" gpurun_out/fused_exp15.log
timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -q 2>&1 | tail -2
