timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
rm -f gpurun_out/fused_exp11.log
for w in pl32768d32 pl32768; do
for s in fuse=1 fuse=1,fuse_meta_nt=1; do
timeout -k 10 120 python tools/leg_only.py --workload $w --steps 5 --set $s >> gpurun_out/fused_exp11.log 2>&1
done; done
cat gpurun_out/fused_exp11.log
