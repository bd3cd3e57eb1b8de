#!/bin/bash
# round 3, batch 5: full -m gpu suite + default bench + 1-rank nccl bench record
set -o pipefail
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/r3e/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3e/tests.log
tail -n 25 gpurun_out/r3e/tests.log
timeout -k 10 400 python bench.py > gpurun_out/r3e/bench_default.json 2> gpurun_out/r3e/bench_default.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3e/bench_default.json').read())
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline'], d['cpu_baseline'])
for k, v in d.get('secondary', {}).items():
    print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'error')}, (v.get('roofline') or {}).get('frac'), v.get('cpu_baseline'))
PY
timeout -k 10 300 python bench.py --gpus 1 --force-dist --no-secondary --no-cpu-baseline > gpurun_out/r3e/bench_1rank_nccl.json 2> gpurun_out/r3e/bench_1rank_nccl.err
echo "nccl rc=$?"; tail -n 5 gpurun_out/r3e/bench_1rank_nccl.err; cut -c1-600 gpurun_out/r3e/bench_1rank_nccl.json
