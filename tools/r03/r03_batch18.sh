#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "logpost_grad_matches_oracle or random_specs" > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; tail -n 8 $O/parity.log | cut -c1-220
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log | cut -c1-200
{
for E in 12 128; do echo "== [5,64,64,64,2] tanh regr N=1052 E=$E"; timeout -k 10 100 python tools/shape_time.py 5 64,64,64,2 regr 1052 $E generic,mfma_narrow_f32 10 tanh; done
echo "== [5,64,64,64,2] relu regr N=1052 E=128 (w64 kernels for reference)"; timeout -k 10 100 python tools/shape_time.py 5 64,64,64,2 regr 1052 128 mfma_w64_bf16x3,mfma_narrow_f32 20 relu
echo "== [54,64,64,7] sigmoid class N=50000 E=128"; timeout -k 10 200 python tools/shape_time.py 54 64,64,7 classification 50000 128 generic,mfma_narrow_f32 3 sigmoid
echo "== [54,32,7] sigmoid class N=232404 E=128"; timeout -k 10 200 python tools/shape_time.py 54 32,7 classification 232404 128 mfma_narrow_f32 3 sigmoid
} 2>&1 | grep -v amdgpu > $O/narrow_mid_time.log; cat $O/narrow_mid_time.log
