#!/bin/bash
# round 3: BASELINE config 3 on the reference's protein table -- bf16-operand kernel vs the fp32-faithful
# layer-wise path from the SAME warm-started members and seeds.  usage: protein_b3.sh <outdir under gpurun_out> [var_start var_end]
set -o pipefail
OUT=gpurun_out/$1; mkdir -p $OUT
VS=${2:-0.5}; VE=${3:-0.1}
export TMPDIR=/tmp
rm -rf /tmp/p3 && mkdir -p /tmp/p3
python - "$VS" "$VE" <<'PY'
import sys, yaml
vs, ve = float(sys.argv[1]), float(sys.argv[2])
c = yaml.safe_load(open('experiments/mclmc_protein_b3.yaml'))
c['saving_dir'] = '/tmp/p3/'
c['training']['sampler'].update(desired_energy_var_start=vs, desired_energy_var_end=ve)
yaml.safe_dump(c, open('/tmp/p3/bf16.yaml', 'w'))
c['experiment_name'] = 'mclmc_protein_3x128_e512_f32x3'
c['training']['sampler']['grad_kernel'] = 'auto'
c['training']['warmstart']['warmstart_exp_dir'] = '/tmp/p3/mclmc_protein_3x128_e512_bf16'
yaml.safe_dump(c, open('/tmp/p3/f32.yaml', 'w'))
PY
( timeout -k 10 700 python train.py -c /tmp/p3/bf16.yaml -d 1 2>&1 | grep -v "Epoch" ) > $OUT/bf16_train.log
echo "bf16 train rc=$?" >> $OUT/bf16_train.log; tail -n 6 $OUT/bf16_train.log
grep -c Epoch /tmp/p3/mclmc_protein_3x128_e512_bf16/training.log
timeout -k 10 300 python evaluate.py -e /tmp/p3/mclmc_protein_3x128_e512_bf16 --drop-nonfinite > $OUT/bf16_eval.log 2>&1
cp /tmp/p3/mclmc_protein_3x128_e512_bf16/metrics.json $OUT/bf16_metrics.json
cp /tmp/p3/mclmc_protein_3x128_e512_bf16/warmup_params.txt $OUT/bf16_warmup_params.txt
tail -n 2 $OUT/bf16_eval.log
rm -rf /tmp/p3/mclmc_protein_3x128_e512_bf16/samples
( timeout -k 10 900 python train.py -c /tmp/p3/f32.yaml -d 1 2>&1 | grep -v "Epoch" ) > $OUT/f32_train.log
echo "f32 train rc=$?" >> $OUT/f32_train.log; tail -n 6 $OUT/f32_train.log
timeout -k 10 300 python evaluate.py -e /tmp/p3/mclmc_protein_3x128_e512_f32x3 --drop-nonfinite > $OUT/f32_eval.log 2>&1
cp /tmp/p3/mclmc_protein_3x128_e512_f32x3/metrics.json $OUT/f32_metrics.json
cp /tmp/p3/mclmc_protein_3x128_e512_f32x3/warmup_params.txt $OUT/f32_warmup_params.txt
tail -n 2 $OUT/f32_eval.log
rm -rf /tmp/p3
