#!/bin/bash
# shards after the pad change: parity tests on the shards, then the emulated per-rank tables (config 4 and 5)
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_kernels.py -m gpu -x -q -k "shard or rccl or wire or pad or stage" > gpurun_out/r04_shard_tests.log 2>&1
echo "pytest shards rc $?"; tail -3 gpurun_out/r04_shard_tests.log
timeout -k 10 400 python tools/emulate_shards.py pl32768d32 > gpurun_out/r04_shards_emulated_pl32768d32.log 2>&1 && \
PS=1,8 timeout -k 10 400 python tools/emulate_shards.py pl65536 > gpurun_out/r04_shards_emulated_pl65536.log 2>&1 && \
timeout -k 10 200 python tools/wire_probe.py > gpurun_out/r04_wire_probe.log 2>&1
cat gpurun_out/r04_shards_emulated_pl32768d32.log gpurun_out/r04_shards_emulated_pl65536.log | cut -c1-250
