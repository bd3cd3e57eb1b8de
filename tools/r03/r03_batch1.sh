#!/bin/bash
# round 3, batch 1: dead-chain trace on B2 (stock targets) + the reference's verbatim stock config through train.py
set -o pipefail
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout -k 10 500 python tools/r03/dead_chain_trace.py experiments/mclmc_airfoil_b2.yaml gpurun_out/r3a/trace_b2.json 3 > gpurun_out/r3a/trace_b2.log 2>&1
echo "trace rc=$?" >> gpurun_out/r3a/trace_b2.log
tail -n 60 gpurun_out/r3a/trace_b2.log
rm -rf /tmp/stock_run && mkdir -p /tmp/stock_run
python - <<'PY'
import yaml
c = yaml.safe_load(open('experiments/mclmc_airfoil_stock.yaml'))
c['saving_dir'] = '/tmp/stock_run/'
yaml.safe_dump(c, open('/tmp/stock_run/cfg.yaml', 'w'))
PY
timeout -k 10 400 python train.py -c /tmp/stock_run/cfg.yaml -d 1 > gpurun_out/r3a/stock_train.log 2>&1
echo "train rc=$?" >> gpurun_out/r3a/stock_train.log
timeout -k 10 200 python evaluate.py -e /tmp/stock_run/mclmc_airfoil_stock_16x16_e12 --drop-nonfinite > gpurun_out/r3a/stock_eval.log 2>&1
cp /tmp/stock_run/mclmc_airfoil_stock_16x16_e12/metrics.json gpurun_out/r3a/stock_metrics.json
cp /tmp/stock_run/mclmc_airfoil_stock_16x16_e12/warmup_params.txt gpurun_out/r3a/stock_warmup_params.txt
cp /tmp/stock_run/mclmc_airfoil_stock_16x16_e12/training.log gpurun_out/r3a/stock_training.log
tail -n 5 gpurun_out/r3a/stock_train.log gpurun_out/r3a/stock_eval.log
cat gpurun_out/r3a/stock_warmup_params.txt
