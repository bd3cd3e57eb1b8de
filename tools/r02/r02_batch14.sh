#!/bin/bash
# rocprofv3 kernel stats of the final tree: B3 (k_update_big), B4 (FULL tile form), B5 / LeNet
O=$GRAFT_REPO_ROOT/gpurun_out/r2p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
echo "[1] B3"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b3 -o b3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload B3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/b3.log 2>&1; echo rc=$?
echo "[2] B4"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b4 -o b4 -- python3 $GRAFT_REPO_ROOT/tools/b4_time.py > $O/b4.log 2>&1; echo rc=$?; tail -1 $O/b4.log
echo "[3] LeNet"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ln -o ln -- python3 $GRAFT_REPO_ROOT/tools/lenet_time.py lenet_bf16 > $O/ln.log 2>&1; echo rc=$?; tail -1 $O/ln.log
echo "[4] B2"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2 -o b2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --no-kernel-timing > $O/b2.log 2>&1; echo rc=$?
echo "[5] done"
