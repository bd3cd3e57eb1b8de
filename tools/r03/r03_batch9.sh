#!/bin/bash
# round 3, batch 9: whole -m gpu suite; phase-3 timing; protein stock with per-chain metrics
set -o pipefail
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r3j/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3j/tests.log; tail -n 16 gpurun_out/r3j/tests.log
timeout -k 10 200 python tools/phase3_time.py > gpurun_out/r3j/phase3_time.log 2>&1; tail -n 8 gpurun_out/r3j/phase3_time.log
bash tools/r03/protein_b3.sh r3j/protein_stock 0.5 0.1
