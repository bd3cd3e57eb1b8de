#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2m; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] lenet tests"; timeout -k 10 900 python -m pytest tests/test_gpu_lenet.py -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log | cut -c1-400; exit 1; }; tail -3 $O/t1.log
echo "[2] timing"; timeout -k 10 300 python tools/lenet_time.py lenet_f32 > $O/time_f32.log 2>&1; tail -1 $O/time_f32.log; timeout -k 10 300 python tools/lenet_time.py lenet_bf16 > $O/time_bf16.log 2>&1; tail -1 $O/time_bf16.log
echo "[3] rocprof f32"; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ln -- python3 $GRAFT_REPO_ROOT/tools/lenet_time.py lenet_f32 > $O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
echo "[4] done"
