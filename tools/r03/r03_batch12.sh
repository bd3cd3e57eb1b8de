#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
{ timeout -k 10 120 python tools/r03/stock_step_time.py 12 4000; timeout -k 10 120 python tools/r03/stock_step_time.py 128 4000; } 2>&1 | grep -v amdgpu > $O/stock_step_time.txt; cat $O/stock_step_time.txt
{
for E in 12 128 1024; do echo "== [5,16,16,2] relu regr N=1052 E=$E"; timeout -k 10 120 python tools/shape_time.py 5 16,16,2 regr 1052 $E generic,mfma_narrow_f32 50; done
for E in 12 128; do echo "== [9,16,16,16,2] relu regr N=36000 E=$E"; timeout -k 10 120 python tools/shape_time.py 9 16,16,16,2 regr 36000 $E generic,mfma_narrow_f32 10; done
for E in 12 128; do echo "== [54,32,7] sigmoid class N=232404 E=$E"; timeout -k 10 200 python tools/shape_time.py 54 32,7 classification 232404 $E generic,mfma_narrow_f32 3 sigmoid; done
} 2>&1 | grep -v amdgpu > $O/narrow_time.log; cat $O/narrow_time.log
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -o ss -- python3 $GRAFT_REPO_ROOT/tools/r03/stock_step_time.py 12 2000 > $O/ss.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
f=$(find $O/ss -name "*kernel_stats.csv" | head -1); cp $f $O/stock_step_kernel_stats.csv; rm -rf $O/ss
head -6 $O/stock_step_kernel_stats.csv | cut -c1-140
rm -rf /tmp/stock_run && mkdir -p /tmp/stock_run
python - <<'PY'
import yaml
c = yaml.safe_load(open('experiments/mclmc_airfoil_stock.yaml'))
c['saving_dir'] = '/tmp/stock_run/'
yaml.safe_dump(c, open('/tmp/stock_run/stock.yaml', 'w'))
PY
( timeout -k 10 400 python train.py -c /tmp/stock_run/stock.yaml -d 1 2>&1 | grep -v "Epoch\|Starting Training" ) | tail -n 5 > $O/stock_train_tail.log; cat $O/stock_train_tail.log
timeout -k 10 200 python evaluate.py -e /tmp/stock_run/mclmc_airfoil_stock_16x16_e12 > $O/stock_eval.log 2>&1; tail -n 1 $O/stock_eval.log | cut -c1-600
cp /tmp/stock_run/mclmc_airfoil_stock_16x16_e12/metrics.json $O/stock_metrics.json
