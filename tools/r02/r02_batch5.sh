#!/bin/bash
# dot2-based three-term split: full GPU suite, error level, B2 / B4 timings
O=$GRAFT_REPO_ROOT/gpurun_out/r2g; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] full gpu suite"; timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/t1.log 2>&1 || { tail -30 $O/t1.log | cut -c1-300; exit 1; }; tail -3 $O/t1.log
echo "[2] split_check"; timeout -k 10 200 python tools/split_check.py > $O/split.log 2>&1; cat $O/split.log
echo "[3] bench B2"; timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }; cut -c1-1500 $O/bench.json
echo "[4] bench B2 steps 20"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-secondary > $O/bench20.json 2> $O/bench20.err; cut -c1-400 $O/bench20.json
echo "[5] B4"; timeout -k 10 300 python tools/b4_time.py > $O/b4.log 2>&1; tail -5 $O/b4.log
echo "[6] done"
