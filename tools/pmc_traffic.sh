#!/bin/bash
# HBM traffic counters, each in its own --pmc pass (TCC slots), kernel-trace only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift   # usage (on the GPU box): tools/pmc_*.sh <outdir> [bench args]
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/t$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/t$i.log 2>&1 || { echo "pass $i ($P) failed"; tail -3 $OUT/t$i.log; }
  i=$((i+1))
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/t*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if 'k_grad' not in k and 'k_update' not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:24s} n={len(v):4d} mean={sum(v)/len(v):16.1f}')
PY
