timeout -k 10 1100 python tools/soak_fits.py 400 2000 > gpurun_out/r03_soak_fits.log 2>&1; echo "soak rc $?"
tail -3 gpurun_out/r03_soak_fits.log
