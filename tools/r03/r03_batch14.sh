#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3o; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log | cut -c1-200
{
echo "== [5,16x4,2] relu regr N=1052 E=12"; timeout -k 10 100 python tools/shape_time.py 5 16,16,16,16,2 regr 1052 12 generic,mfma_narrow_f32 30
echo "== [5,16x9,2] relu regr N=1052 E=12"; timeout -k 10 100 python tools/shape_time.py 5 16,16,16,16,16,16,16,16,16,2 regr 1052 12 generic,mfma_narrow_f32 30
echo "== [8,8x6,2] relu regr N=1052 E=128"; timeout -k 10 100 python tools/shape_time.py 8 8,8,8,8,8,8,2 regr 1052 128 generic,mfma_narrow_f32 30
} 2>&1 | grep -v amdgpu > $O/narrow_deep_time.log; cat $O/narrow_deep_time.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3o/bench_default.json').read())
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'])
for k, v in d.get('secondary', {}).items():
    print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'error')}, (v.get('roofline') or {}).get('frac'), (v.get('cpu_baseline') or {}).get('value'))
PY
bash tools/r03/pmc_whole_gradient.sh r3o/b5_traffic k_conv5m_fwd2x --workload B5 --steps 4 --warmup 1 > $O/b5_traffic.txt 2>&1; tail -n 25 $O/b5_traffic.txt
