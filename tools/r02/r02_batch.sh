#!/bin/bash
# round-2 measurement batch (run on the GPU box): tests, bench lines, rocprof summaries, counters
O=$GRAFT_REPO_ROOT/gpurun_out/r2d; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] pytest"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -2 $O/tests.log
echo "[2] bench default"; timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
echo "[3] bench --steps 20"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-secondary > $O/bench_steps20.json 2>/dev/null; cut -c1-200 $O/bench_steps20.json
echo "[4] bench B4"; timeout -k 10 900 python bench.py --workload B4 > $O/bench_b4.json 2> $O/bench_b4.err; cut -c1-300 $O/bench_b4.json
echo "[5] rocprof stats B2"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --no-kernel-timing > $O/prof_b2.log 2>&1); find $O/prof_b2 -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-160
echo "[6] rocprof stats B3"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload B3 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/prof_b3.log 2>&1); find $O/prof_b3 -name "*kernel_stats.csv" | head -1 | xargs head -5 | cut -c1-160
echo "[7] pmc traffic B2"; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_traffic -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-kernel-timing > $O/pmc_traffic.log 2>&1); echo done
echo "[8] 2-rank rehearsal"; timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err; cut -c1-200 $O/bench_2rank_rehearsal.json; grep "rank" $O/bench_2rank_rehearsal.err | head -4
echo "[9] airfoil stock, drop-nonfinite"; python train.py -c experiments/mclmc_airfoil_b2.yaml --silent > $O/airfoil_stock.train.log 2>&1; D=results/mile_amd/mclmc_airfoil_3x64_e128; grep "time\.\|Warmup sampling completed\|stepping" $D/training.log | cut -c1-200 > $O/airfoil_stock.times.log; python evaluate.py -e $D --drop-nonfinite > $O/airfoil_stock.metrics_dropnonfinite.json 2> $O/airfoil_stock.eval.err; cat $O/airfoil_stock.metrics_dropnonfinite.json
echo "[10] done"
