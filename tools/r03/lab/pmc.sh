#!/bin/bash
# SQ counters of one lab binary (separate --pmc passes, kernel-trace only): tools/r03/lab/pmc.sh <binary> [args]
B=$GRAFT_REPO_ROOT/tools/r03/lab/bin/$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/lab_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES"
P3="GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_CYCLES SQ_ACTIVE_INST_MISC"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- $B 512 36000 3 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  i=$((i+1))
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:64]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}')
PY
