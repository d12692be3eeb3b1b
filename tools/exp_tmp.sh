rm -f gpurun_out/fused_exp10.log
B=fuse=1,fuse_min=3,fuse_unit=1000000
for w in pl32768d32; do
for s in $B $B,fuse_order=1 $B,fuse_order=2 $B,fuse_order=4 $B,fuse_order=1,fuse_min=2 $B,fuse_order=2,fuse_min=2 $B,fuse_order=1,probe_mask=255 $B,fuse_order=2,fuse_unit=64 $B,fuse_order=2,fuse_unit=32; do
timeout -k 10 120 python tools/leg_only.py --workload $w --steps 5 --set $s >> gpurun_out/fused_exp10.log 2>&1
done; done
cat gpurun_out/fused_exp10.log
