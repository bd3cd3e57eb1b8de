#!/bin/bash
# the round-end checks: build, smoke, full GPU suite
O=$GRAFT_REPO_ROOT/gpurun_out/r2full; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] smoke"; timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }; tail -2 $O/smoke.log
echo "[2] full gpu suite"; timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=8 > $O/t.log 2>&1 || { tail -40 $O/t.log | cut -c1-300; exit 1; }; tail -14 $O/t.log
echo "[3] done"
