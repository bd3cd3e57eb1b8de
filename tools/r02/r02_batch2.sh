#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2d; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[7] pmc traffic B2 (separate passes)"
for P in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/traffic_$P -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-kernel-timing > $O/traffic_$P.log 2>&1); echo "  $P rc=$?"
done
python3 - $O <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/traffic_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if 'k_grad' not in k and 'k_update' not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:24s} n={len(v):4d} mean={sum(v)/len(v):16.1f}')
PY
echo "[8] 2-rank rehearsal"; timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err; cut -c1-200 $O/bench_2rank_rehearsal.json; grep "rank" $O/bench_2rank_rehearsal.err | head -4
echo "[9] airfoil stock, drop-nonfinite"; python train.py -c experiments/mclmc_airfoil_b2.yaml --silent > $O/airfoil_stock.train.log 2>&1; D=results/mile_amd/mclmc_airfoil_3x64_e128; grep "time\.\|Warmup sampling completed\|stepping" $D/training.log | cut -c1-200 > $O/airfoil_stock.times.log; echo "  trained"; python evaluate.py -e $D --drop-nonfinite > $O/airfoil_stock.metrics_dropnonfinite.json 2> $O/airfoil_stock.eval.err; cat $O/airfoil_stock.metrics_dropnonfinite.json
echo "[10] done"
