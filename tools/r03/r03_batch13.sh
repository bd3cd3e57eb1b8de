#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log | cut -c1-200
{ timeout -k 10 120 python tools/r03/stock_step_time.py 12 4000; timeout -k 10 120 python tools/r03/stock_step_time.py 128 4000; } 2>&1 | grep -v amdgpu > $O/stock_step_time.txt; cat $O/stock_step_time.txt
{
for E in 12 128; do echo "== [5,16,16,2] relu regr N=1052 E=$E"; timeout -k 10 120 python tools/shape_time.py 5 16,16,2 regr 1052 $E mfma_narrow_f32 50; done
echo "== [9,16,16,16,2] N=36000 E=12"; timeout -k 10 120 python tools/shape_time.py 9 16,16,16,2 regr 36000 12 mfma_narrow_f32 10
echo "== [54,32,7] sigmoid class N=232404 E=12"; timeout -k 10 200 python tools/shape_time.py 54 32,7 classification 232404 12 mfma_narrow_f32 3 sigmoid
} 2>&1 | grep -v amdgpu > $O/narrow_time.log; cat $O/narrow_time.log
rm -rf /tmp/run && mkdir -p /tmp/run
python - <<'PY'
import yaml
for n in ('airfoil_stock', 'airfoil_b2', 'protein_b3'):
    c = yaml.safe_load(open('experiments/mclmc_%s.yaml' % n))
    c['saving_dir'] = '/tmp/run/'
    if n == 'protein_b3':
        c['training']['sampler'].update(warmup_steps=100, n_samples=10)
    yaml.safe_dump(c, open('/tmp/run/%s.yaml' % n, 'w'))
PY
for n in airfoil_stock airfoil_b2 protein_b3; do ( timeout -k 10 500 python train.py -c /tmp/run/$n.yaml -d 1 2>&1 | grep -v "Epoch\|Starting Training" ) | grep "took\|completed\|stepping" > $O/${n}_train_tail.log; cat $O/${n}_train_tail.log | cut -c1-200; done
for e in mclmc_airfoil_stock_16x16_e12 mclmc_airfoil_3x64_e128; do timeout -k 10 200 python evaluate.py -e /tmp/run/$e --drop-nonfinite 2>&1 | tail -n 1 | cut -c1-500; cp /tmp/run/$e/metrics.json $O/${e}_metrics.json; done
