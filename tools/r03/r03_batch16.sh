#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -n 3
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log | cut -c1-200
MILE_DEBUG=16 timeout -k 10 200 python tools/b3_time.py 512 36000 mfma_w128_bf16 2>&1 | grep -v amdgpu > $O/b3_phase_stamps.txt; tail -n 6 $O/b3_phase_stamps.txt | cut -c1-250
bash tools/pmc_script.sh r3q/b3_sq $GRAFT_REPO_ROOT/tools/b3_time.py 512 36000 mfma_w128_bf16 > $O/b3_sq.txt 2>&1; grep -A26 "k_grad_w128b" $O/b3_sq.txt | head -30
rm -rf $O/b3_sq/p*/ 2>/dev/null
