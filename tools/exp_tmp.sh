timeout -k 10 1100 python tools/soak_plan.py 0 600 > gpurun_out/r03_soak_plan.log 2>&1; echo "soak rc $?"
grep "soak_plan\|Error\|assert" gpurun_out/r03_soak_plan.log | tail -5
