#!/bin/bash
# round 3, batch 2: dead-chain trace again (with the guarded gradient normalisation + nan_to_num on every leaf) + fixture,
# B2 stock schedule end to end after the fix (dead chains, LPPD), the new aliasing tests
set -o pipefail
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
timeout -k 10 500 python tools/r03/dead_chain_trace.py experiments/mclmc_airfoil_b2.yaml gpurun_out/r3b/trace_b2.json 3 > gpurun_out/r3b/trace_b2.log 2>&1
echo "trace rc=$?" >> gpurun_out/r3b/trace_b2.log
grep -v "^{'step'" gpurun_out/r3b/trace_b2.log | tail -n 12
rm -rf /tmp/b2_run && mkdir -p /tmp/b2_run
python - <<'PY'
import yaml
c = yaml.safe_load(open('experiments/mclmc_airfoil_b2.yaml'))
c['saving_dir'] = '/tmp/b2_run/'
yaml.safe_dump(c, open('/tmp/b2_run/cfg.yaml', 'w'))
PY
timeout -k 10 400 python train.py -c /tmp/b2_run/cfg.yaml -d 1 > gpurun_out/r3b/b2_train.log 2>&1
echo "train rc=$?" >> gpurun_out/r3b/b2_train.log
timeout -k 10 300 python evaluate.py -e /tmp/b2_run/mclmc_airfoil_3x64_e128 --drop-nonfinite > gpurun_out/r3b/b2_eval.log 2>&1
cp /tmp/b2_run/mclmc_airfoil_3x64_e128/metrics.json gpurun_out/r3b/b2_metrics.json
cp /tmp/b2_run/mclmc_airfoil_3x64_e128/warmup_params.txt gpurun_out/r3b/b2_warmup_params.txt
grep -v Epoch /tmp/b2_run/mclmc_airfoil_3x64_e128/training.log > gpurun_out/r3b/b2_training.log
tail -n 3 gpurun_out/r3b/b2_train.log gpurun_out/r3b/b2_eval.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "alias or tuner or golden or nonfinite or sharding" > gpurun_out/r3b/tests.log 2>&1
tail -n 5 gpurun_out/r3b/tests.log
