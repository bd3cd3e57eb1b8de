#!/bin/bash
# round 3, batch 4: merged warm-up launch -- tests, timing against the five-launch form, kernel trace
set -o pipefail
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "tuner or merged or dead_chain or phase3 or warmup or inference_loop or alias or train_cli" > gpurun_out/r3d/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3d/tests.log
tail -n 30 gpurun_out/r3d/tests.log
timeout -k 10 300 python tools/r03/tune_time.py 2048 > gpurun_out/r3d/tune_time_merged.log 2>&1
MILE_TUNE_NO_MERGE=1 timeout -k 10 300 python tools/r03/tune_time.py 2048 > gpurun_out/r3d/tune_time_five_launch.log 2>&1
cat gpurun_out/r3d/tune_time_merged.log gpurun_out/r3d/tune_time_five_launch.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3d/prof -- python3 $GRAFT_REPO_ROOT/tools/r03/tune_time.py 1024 > $GRAFT_REPO_ROOT/gpurun_out/r3d/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3d/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3d/tune_kernel_stats.csv
head -12 gpurun_out/r3d/tune_kernel_stats.csv | cut -c1-220
rm -rf gpurun_out/r3d/prof
