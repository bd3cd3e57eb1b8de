#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2k; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] wide tests"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -q -x -k "wide or b4 or classification" > $O/t1.log 2>&1 || { tail -30 $O/t1.log | cut -c1-300; exit 1; }; tail -2 $O/t1.log
echo "[2] B4 time FULL"; timeout -k 10 300 python tools/b4_time.py > $O/b4_full.log 2>&1; tail -2 $O/b4_full.log
echo "[3] B4 time without the head block"; MILE_WIDE_NO_HEADBLOCK=1 timeout -k 10 300 python tools/b4_time.py > $O/b4_gen.log 2>&1; tail -2 $O/b4_gen.log
echo "[4] done"
