#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2e; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] window + warmstart tests"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -q -x -k "row_window or warmstart" > $O/t1.log 2>&1; tail -3 $O/t1.log
echo "[2] ws probe"; timeout -k 10 300 python tools/ws_probe.py > $O/ws.log 2>&1; tail -8 $O/ws.log
for c in mclmc_airfoil_b2 mclmc_airfoil_b2_tight; do
  echo "[3] $c"; python train.py -c experiments/$c.yaml --silent > $O/$c.train.log 2>&1; D=$(ls -d results/mile_amd/mclmc_airfoil_3x64_e128* | tail -1)
  grep "Epoch\|Warmstart Training completed" $D/training.log | tail -3 | cut -c60-200
  grep "time\.\|Warmup sampling completed\|stepping" $D/training.log | cut -c1-200 > $O/$c.times.log
  python evaluate.py -e $D --drop-nonfinite > $O/$c.metrics_dropnonfinite.json 2> $O/$c.eval.err; cat $O/$c.metrics_dropnonfinite.json
  mv $D $D.done
done
echo "[4] done"
