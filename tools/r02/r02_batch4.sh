#!/bin/bash
# k_update_big: parity vs the two-pass form, B3 bench before/after, kernel stats
O=$GRAFT_REPO_ROOT/gpurun_out/r2f; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] tests"; timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "register_cache or properties_b3 or post_kernel" > $O/t1.log 2>&1 || { tail -30 $O/t1.log; exit 1; }; tail -3 $O/t1.log
echo "[2] B3 bench, two-pass update"; MILE_NO_UPD_BIG=1 timeout -k 10 300 python bench.py --workload B3 --steps 20 --warmup 5 > $O/b3_twopass.json 2> $O/b3_twopass.err || { tail $O/b3_twopass.err; exit 1; }; cat $O/b3_twopass.json
echo "[3] B3 bench, k_update_big"; timeout -k 10 300 python bench.py --workload B3 --steps 20 --warmup 5 > $O/b3_big.json 2> $O/b3_big.err || { tail $O/b3_big.err; exit 1; }; cat $O/b3_big.json
echo "[4] rocprof B3"; cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload B3 --steps 20 --warmup 5 > $O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 $f | head -8
echo "[5] done"
