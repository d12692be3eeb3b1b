ONLY_HALF=1 bash tools/gpu_profile.sh r03 > gpurun_out/prof_r03_half.txt 2>&1; echo "profile rc $?"
grep -n "config 5" -A8 gpurun_out/prof_r03_half.txt | head -30
grep "^half_" gpurun_out/prof_r03_half.txt
