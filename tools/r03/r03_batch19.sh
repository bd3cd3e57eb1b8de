#!/bin/bash
# round 3, batch 19: k_grad_w128b on padded LDS images -- parity, B3 time, phase stamps
set -o pipefail
mkdir -p gpurun_out/b19
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16_w128" > gpurun_out/b19/pytest_bf16.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/b19/pytest_bf16.txt
tail -3 gpurun_out/b19/pytest_bf16.txt
timeout -k 10 200 python tools/b3_time.py 512 36000 mfma_w128_bf16 > gpurun_out/b19/b3_time.txt 2>&1
cat gpurun_out/b19/b3_time.txt
MILE_DEBUG=16 timeout -k 10 200 python tools/b3_time.py 512 36000 mfma_w128_bf16 > gpurun_out/b19/b3_stamps.txt 2>&1
grep -v "^w128b cycles" gpurun_out/b19/b3_stamps.txt; grep "^w128b cycles" gpurun_out/b19/b3_stamps.txt | head -5
