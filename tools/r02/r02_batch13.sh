#!/bin/bash
# final tree: LeNet tests (odd-width pair case), SQ counters of the B2 kernels of this round
O=$GRAFT_REPO_ROOT/gpurun_out/r2o; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] lenet tests"; timeout -k 10 900 python -m pytest tests/test_gpu_lenet.py -m gpu -q -x > $O/t1.log 2>&1 || { tail -40 $O/t1.log | cut -c1-400; exit 1; }; tail -2 $O/t1.log
echo "[2] SQ counters B2"; bash tools/pmc_counters.sh r2o/pmc --no-secondary > $O/pmc.txt 2>&1; tail -60 $O/pmc.txt | cut -c1-120
echo "[3] done"
