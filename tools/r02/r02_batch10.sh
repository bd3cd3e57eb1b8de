#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2l; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] window + warmstart + lenet cli"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_lenet.py -m gpu -q -x -k "row_window or warmstart or cli" > $O/t1.log 2>&1 || { tail -40 $O/t1.log | cut -c1-400; exit 1; }; tail -3 $O/t1.log
echo "[2] done"
