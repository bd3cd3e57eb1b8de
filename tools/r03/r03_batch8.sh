#!/bin/bash
# round 3, batch 8: whole -m gpu suite with the narrow kernel behind AUTO; stock YAML end to end; protein stock with per-chain metrics
set -o pipefail
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r3i/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3i/tests.log; tail -n 16 gpurun_out/r3i/tests.log
rm -rf /tmp/stock_run && mkdir -p /tmp/stock_run
python - <<'PY'
import yaml
c = yaml.safe_load(open('experiments/mclmc_airfoil_stock.yaml'))
c['saving_dir'] = '/tmp/stock_run/'
yaml.safe_dump(c, open('/tmp/stock_run/cfg.yaml', 'w'))
PY
( timeout -k 10 400 python train.py -c /tmp/stock_run/cfg.yaml -d 1 2>&1 | grep -v Epoch ) > gpurun_out/r3i/stock_train.log
timeout -k 10 200 python evaluate.py -e /tmp/stock_run/mclmc_airfoil_stock_16x16_e12 --drop-nonfinite > gpurun_out/r3i/stock_eval.log 2>&1
cp /tmp/stock_run/mclmc_airfoil_stock_16x16_e12/metrics.json gpurun_out/r3i/stock_metrics.json
tail -n 6 gpurun_out/r3i/stock_train.log; tail -n 1 gpurun_out/r3i/stock_eval.log | cut -c1-400
rm -rf /tmp/b2_run && mkdir -p /tmp/b2_run
python - <<'PY'
import yaml
c = yaml.safe_load(open('experiments/mclmc_airfoil_b2.yaml'))
c['saving_dir'] = '/tmp/b2_run/'
yaml.safe_dump(c, open('/tmp/b2_run/cfg.yaml', 'w'))
PY
( timeout -k 10 400 python train.py -c /tmp/b2_run/cfg.yaml -d 1 2>&1 | grep -v Epoch ) > gpurun_out/r3i/b2_train.log
tail -n 6 gpurun_out/r3i/b2_train.log
