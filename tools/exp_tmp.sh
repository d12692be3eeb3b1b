timeout -k 10 500 python tools/soak_kernels.py 2000 7 > gpurun_out/r03_soak2.log 2>&1; echo "soak rc $?"; tail -1 gpurun_out/r03_soak2.log
timeout -k 10 500 python tools/soak_fits_half.py 600 900 > gpurun_out/r03_soak_fits_half2.log 2>&1; echo "soak rc $?"; tail -1 gpurun_out/r03_soak_fits_half2.log
