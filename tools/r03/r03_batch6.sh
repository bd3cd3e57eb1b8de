#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3f
timeout -k 10 300 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "full_size_properties_b2 or warmstart or dead_chain or merged" > gpurun_out/r3f/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3f/tests.log; tail -n 8 gpurun_out/r3f/tests.log
bash tools/r03/protein_b3.sh r3f/protein_stock 0.5 0.1
