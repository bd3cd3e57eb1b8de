#!/bin/bash
# round 3, batch 7: the narrow-net MFMA kernel -- parity cases, timing against the generic kernel, the suite
set -o pipefail
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "logpost_grad_matches_oracle" > gpurun_out/r3h/parity.log 2>&1
echo "rc=$?" >> gpurun_out/r3h/parity.log; tail -n 15 gpurun_out/r3h/parity.log
{
for E in 12 128 1024; do echo "== [5,16,16,2] relu regr N=1052 E=$E"; timeout -k 10 120 python tools/shape_time.py 5 16,16,2 regr 1052 $E generic,mfma_narrow_f32 50; done
for E in 12 128; do echo "== [9,16,16,16,2] relu regr N=36000 E=$E"; timeout -k 10 120 python tools/shape_time.py 9 16,16,16,2 regr 36000 $E generic,mfma_narrow_f32 10; done
for E in 12 128; do echo "== [54,32,7] sigmoid class N=232404 E=$E"; timeout -k 10 200 python tools/shape_time.py 54 32,7 classification 232404 $E generic,mfma_narrow_f32 3 sigmoid; done
} > gpurun_out/r3h/narrow_time.log 2>&1
grep -v amdgpu gpurun_out/r3h/narrow_time.log
