#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2n; mkdir -p $O; cd $GRAFT_REPO_ROOT
echo "[1] default bench"; s=$(date +%s); timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }; echo "took $(( $(date +%s) - s )) s"; python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2n/bench.json'))
print('B2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])
for k,v in d['secondary'].items(): print(k, v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline']['avg_launch_us'], v['cpu_baseline'] and round(v['cpu_baseline']['value'],3))
PY
echo "[2] driver-style"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; cut -c1-200 $O/bench20.json
echo "[3] B4"; timeout -k 10 600 python bench.py --workload B4 > $O/bench_b4.json 2> $O/bench_b4.err; cut -c1-200 $O/bench_b4.json
echo "[4] B5"; timeout -k 10 600 python bench.py --workload B5 > $O/bench_b5.json 2> $O/bench_b5.err; cut -c1-200 $O/bench_b5.json
echo "[5] 2-rank rehearsal on one GPU"; timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err; cut -c1-260 $O/bench_2rank.json; grep "rank" $O/bench_2rank.err | head -3
echo "[6] done"
