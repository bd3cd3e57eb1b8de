#!/bin/bash
# HBM traffic of a WHOLE gradient of a multi-kernel path (LeNet / B5, k_mm3 / B4): FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes (kernel-trace only), summed over every dispatch and divided by the number of gradient evaluations (= dispatches of the
# kernel named in $2, which runs once per gradient).  usage (on the GPU box): pmc_whole_gradient.sh <outdir> <once-per-gradient kernel> <bench args...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ONCE=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/t$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-secondary "$@" > $OUT/t$i.log 2>&1 || { echo "pass $i ($P) failed"; tail -3 $OUT/t$i.log; }
  i=$((i+1))
done
python3 - $OUT "$ONCE" <<'PY'
import sys, glob, csv, collections, json
out, once = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n_once = collections.defaultdict(int); per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/t*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        c = r['Counter_Name']; k = r['Kernel_Name'].split('(')[0][:48]
        if not k.startswith(('void k_', 'k_')): continue            # library kernels only (torch fills / copies of the harness left out)
        tot[c] += float(r['Counter_Value']); per_kernel[k][c] += float(r['Counter_Value'])
        if once in k: n_once[c] += 1
n = max(n_once.values()) if n_once else 0
res = {'gradient_evaluations': n, 'once_per_gradient_kernel': once}
if n:
    f_kib, w_kib = tot['FETCH_SIZE'] / n, tot['WRITE_SIZE'] / n
    res.update(fetch_size_kib_per_gradient=f_kib, write_size_kib_per_gradient=w_kib, fetch_correction=2.0,
               hbm_bytes_per_gradient=(2.0 * f_kib + w_kib) * 1024)
    res['per_kernel_gb'] = {k: round((2.0 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) * 1024 / n / 1e9, 3) for k, v in per_kernel.items()}
print(json.dumps(res, indent=1))
open(out + '/whole_gradient_traffic.json', 'w').write(json.dumps(res, indent=1))
PY
rm -rf $OUT/t1 $OUT/t2
