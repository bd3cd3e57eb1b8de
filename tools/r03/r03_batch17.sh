#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3r; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log | cut -c1-200
{
for E in 12 128; do echo "== [54,32,7] sigmoid class N=232404 E=$E"; timeout -k 10 200 python tools/shape_time.py 54 32,7 classification 232404 $E generic,mfma_narrow_f32 3 sigmoid; done
echo "== [11,32,5] sigmoid class N=4000 E=12"; timeout -k 10 100 python tools/shape_time.py 11 32,5 classification 4000 12 generic,mfma_narrow_f32 20 sigmoid
} 2>&1 | grep -v amdgpu > $O/narrow_sigmoid_time.log; cat $O/narrow_sigmoid_time.log
