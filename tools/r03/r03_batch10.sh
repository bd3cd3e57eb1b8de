#!/bin/bash
# round 3, batch 10: profiles of the final tree (rocprofv3 kernel stats, PMC traffic, narrow-kernel SQ counters), FFT JIT probe,
# protein equal-steps comparison
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp
for v in a b c d; do timeout -k 10 120 python tools/r03/fft_jit_probe.py $v 2>&1 | grep -v amdgpu; done > $O/fft_jit_probe.txt; cat $O/fft_jit_probe.txt
cd /tmp
echo "[1] B2 kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2 -o b2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --no-kernel-timing > $O/b2.log 2>&1; echo rc=$?
echo "[2] B3 kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b3 -o b3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload B3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/b3.log 2>&1; echo rc=$?
echo "[3] stock-net step kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- python3 $GRAFT_REPO_ROOT/tools/shape_time.py 5 16,16,2 regr 1052 128 mfma_narrow_f32 200 > $O/st.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
for n in b2 b3 st; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv; rm -rf $O/$n; done
head -6 $O/b2_kernel_stats.csv | cut -c1-160
echo "[4] PMC traffic B2"; bash tools/pmc_traffic.sh r3k/traffic > $O/pmc_traffic.txt 2>&1; tail -n 14 $O/pmc_traffic.txt
echo "[5] SQ counters narrow kernel (covertype shape, E=128)"; bash tools/pmc_script.sh r3k/narrow_sq tools/shape_time.py 54 32,7 classification 232404 128 mfma_narrow_f32 3 sigmoid > $O/narrow_sq.txt 2>&1; tail -n 30 $O/narrow_sq.txt
rm -rf $O/traffic/t*/ $O/narrow_sq/p*/ 2>/dev/null
echo "[6] protein equal steps"; timeout -k 10 800 python tools/r03/protein_equal_steps.py gpurun_out/r3k/protein_equal_steps_stock.json 2>&1 | grep -v "Epoch\|Starting Training" | tail -n 3
