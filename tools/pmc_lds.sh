#!/bin/bash
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/lds
cd /tmp
for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/lds/$C -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu-baseline --no-other-configs --profile-steps 0 --no-overlap --rows 262144 > $ROOT/gpurun_out/lds/$C.log 2>&1 || echo "fail $C"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(dict)
for C in ['SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_ACTIVE_INST_LDS','SQ_WAVE_CYCLES','SQ_INSTS_LDS']:
    fs = glob.glob('gpurun_out/lds/%s/*/*counter_collection.csv' % C)
    if not fs: print('no file', C); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r['Counter_Name'] == C: acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        if 'gemm_f64_kernel' in k and len(v) >= 8: res[k][C] = sum(v) / len(v)
for k, v in res.items(): print(k.split('(zigp::GemmArgs')[0].replace('void zigp::', ''), {a: '%.3g' % b for a, b in v.items()})
PY
