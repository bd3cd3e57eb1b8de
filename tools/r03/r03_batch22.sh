#!/bin/bash
# round 3, batch 22: reads-in-gaps form of k_grad_w128b: bf16 parity, B3 kernel stats, lab timing stamps
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/b22; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16_w128" > $O/pytest_bf16.txt 2>&1; rc=$?; tail -2 $O/pytest_bf16.txt
[ $rc -eq 0 ] || exit 1
bash tools/r03/lab/run.sh base timing | tee $O/lab.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b3 -o b3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload B3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/b3.log 2>&1; echo rc=$?
head -3 $O/b3/*kernel_stats.csv; tail -1 $O/b3.log | cut -c1-400
