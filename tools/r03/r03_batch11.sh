#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 120 python tools/r03/stock_step_time.py 12 4000 2>&1 | grep -v amdgpu > $O/stock_step_time.txt
timeout -k 10 120 python tools/r03/stock_step_time.py 128 4000 2>&1 | grep -v amdgpu >> $O/stock_step_time.txt
cat $O/stock_step_time.txt
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -o ss -- python3 $GRAFT_REPO_ROOT/tools/r03/stock_step_time.py 12 2000 > $O/ss.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
f=$(find $O/ss -name "*kernel_stats.csv" | head -1); cp $f $O/stock_step_kernel_stats.csv; rm -rf $O/ss
head -8 $O/stock_step_kernel_stats.csv | cut -c1-140
bash tools/pmc_script.sh r3l/narrow_sq $GRAFT_REPO_ROOT/tools/shape_time.py 54 32,7 classification 232404 128 mfma_narrow_f32 3 sigmoid > $O/narrow_sq.txt 2>&1; tail -n 28 $O/narrow_sq.txt
rm -rf $O/narrow_sq/p*/ 2>/dev/null
rm -rf /tmp/stock_run && mkdir -p /tmp/stock_run
python - <<'PY'
import yaml
for n in ('stock', 'b2'):
    c = yaml.safe_load(open('experiments/mclmc_airfoil_%s.yaml' % n))
    c['saving_dir'] = '/tmp/stock_run/'
    yaml.safe_dump(c, open('/tmp/stock_run/%s.yaml' % n, 'w'))
PY
for n in stock b2; do ( timeout -k 10 400 python train.py -c /tmp/stock_run/$n.yaml -d 1 2>&1 | grep -v "Epoch\|Starting Training" ) | tail -n 5 > $O/${n}_train_tail.log; cat $O/${n}_train_tail.log; done
