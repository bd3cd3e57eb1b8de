#!/bin/bash
# round 3, batch 3: the whole -m gpu suite (new tuner / dead-chain / aliasing tests included)
set -o pipefail
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r3c/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3c/tests.log
tail -n 40 gpurun_out/r3c/tests.log
