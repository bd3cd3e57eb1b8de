#!/bin/bash
# round 3, batch 21: full GPU suite on the tree with padded-image kernels + rocprofv3 kernel stats of B2 and B3
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/b21; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
echo "[1] B2 kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2 -o b2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --no-kernel-timing > $O/b2.log 2>&1; echo rc=$?
echo "[2] B3 kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b3 -o b3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload B3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/b3.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
for f in $O/b2/*kernel_stats.csv $O/b3/*kernel_stats.csv; do echo $f; head -6 $f; done
python -c "import __graft_entry__ as g; g.smoke()"
