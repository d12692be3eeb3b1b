timeout -k 10 1100 python tools/soak_fits_half.py 0 600 > gpurun_out/r03_soak_fits_half.log 2>&1; echo "soak rc $?"
tail -6 gpurun_out/r03_soak_fits_half.log
