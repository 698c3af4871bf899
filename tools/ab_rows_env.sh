#!/bin/bash
# same-box A/B of environment settings at a given row count / M (per-kernel TF from bench.py's profiled pass): tools/ab_rows_env.sh ROWS M ROUNDS "ENV=.." "-" ...
ROWS=$1; M=$2; R=$3; shift 3
for r in $(seq 1 $R); do for V in "$@"; do
  E="$V"; [ "$V" = "-" ] && E=""
  env $E timeout -k 10 300 python bench.py --rows $ROWS --M $M --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); pp=d['profiled_pass']; print('[$V]', round(d['ms_per_step'],3), {k:round(v,1) for k,v in d['roofline']['per_kernel_tflops'].items()}, 'mxm', round(pp['mxm_stage_ms_both_streams'],3), {k:round(v,2) for k,v in pp['kernel_ms_per_step'].items() if v>0})"
done; done
