#!/bin/bash
# round 3, batch 20: k_grad_w64 on padded bf16 images -- parity suite, B2 bench, kernel stats
set -o pipefail
mkdir -p gpurun_out/b20
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/b20/pytest_parity.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/b20/pytest_parity.txt
tail -4 gpurun_out/b20/pytest_parity.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > gpurun_out/b20/bench_default.json 2> gpurun_out/b20/bench_default.err; cat gpurun_out/b20/bench_default.json
