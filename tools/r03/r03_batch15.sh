#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -q -m gpu -k "tuner or merged or dead_chain or warmup or golden or sharding or rejects or fused" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log | cut -c1-200
timeout -k 10 200 python tools/r03/tune_time.py 2048 2>&1 | grep -v amdgpu > $O/tune_time.txt; cat $O/tune_time.txt
cd /tmp; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tt -o tt -- python3 $GRAFT_REPO_ROOT/tools/r03/tune_time.py 1024 > $O/tt.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/tt -name "*kernel_stats.csv" | head -1); cp $f $O/tune_kernel_stats.csv; rm -rf $O/tt; head -7 $O/tune_kernel_stats.csv | cut -c1-140
rm -rf /tmp/run && mkdir -p /tmp/run
python - <<'PY'
import yaml
c = yaml.safe_load(open('experiments/mclmc_airfoil_b2.yaml'))
c['saving_dir'] = '/tmp/run/'
yaml.safe_dump(c, open('/tmp/run/b2.yaml', 'w'))
PY
( timeout -k 10 300 python train.py -c /tmp/run/b2.yaml -d 1 2>&1 | grep -v "Epoch\|Starting Training" ) | grep "took\|completed\|stepping" > $O/b2_train_tail.log; cat $O/b2_train_tail.log | cut -c1-200
