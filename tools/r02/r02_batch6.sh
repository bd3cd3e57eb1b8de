#!/bin/bash
# LeNet convolutions on the matrix pipe: parity, timing vs the fp32 direct kernels, kernel stats
O=$GRAFT_REPO_ROOT/gpurun_out/r2h; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] lenet tests"; timeout -k 10 900 python -m pytest tests/test_gpu_lenet.py -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log | cut -c1-400; exit 1; }; tail -3 $O/t1.log
echo "[2] timing f32"; timeout -k 10 300 python tools/lenet_time.py lenet_f32 > $O/time_f32.log 2>&1; tail -2 $O/time_f32.log
echo "[3] timing bf16"; timeout -k 10 300 python tools/lenet_time.py lenet_bf16 > $O/time_bf16.log 2>&1; tail -2 $O/time_bf16.log
echo "[4] rocprof bf16"; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ln -- python3 $GRAFT_REPO_ROOT/tools/lenet_time.py lenet_bf16 > $O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 $f | head -14
echo "[5] done"
