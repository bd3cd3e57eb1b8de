#!/bin/bash
# LeNet bf16: all lenet tests (with the config-5 full-size property test), timing, profile for profiles/r02
O=$GRAFT_REPO_ROOT/gpurun_out/r2i; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] lenet tests"; timeout -k 10 900 python -m pytest tests/test_gpu_lenet.py -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log | cut -c1-400; exit 1; }; tail -3 $O/t1.log
echo "[2] timing"; timeout -k 10 300 python tools/lenet_time.py lenet_f32 > $O/time_f32.log 2>&1; tail -1 $O/time_f32.log; timeout -k 10 300 python tools/lenet_time.py lenet_bf16 > $O/time_bf16.log 2>&1; tail -1 $O/time_bf16.log
echo "[3] ipw 16"; MILE_LENET_IPW=16 timeout -k 10 300 python tools/lenet_time.py lenet_bf16 > $O/time_bf16_ipw16.log 2>&1; tail -1 $O/time_bf16_ipw16.log
echo "[4] ipw 4"; MILE_LENET_IPW=4 timeout -k 10 300 python tools/lenet_time.py lenet_bf16 > $O/time_bf16_ipw4.log 2>&1; tail -1 $O/time_bf16_ipw4.log
echo "[5] rocprof bf16"; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ln -- python3 $GRAFT_REPO_ROOT/tools/lenet_time.py lenet_bf16 > $O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
echo "[6] done"
