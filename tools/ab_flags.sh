#!/bin/bash
# A/B of bench.py FLAG SETS of one build on the SAME GPU box, interleaved rounds: tools/ab_flags.sh ROUNDS "flags A" "flags B" ...
R=$1; shift
for r in $(seq 1 $R); do for F in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-other-configs $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$F]', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['roofline']['per_kernel_tflops'].items()}, {k:round(v,2) for k,v in d['profiled_pass']['kernel_ms_per_step'].items() if k in ('kgrad','kuf_build','pointwise','syrk')}, 'mxm', round(d['profiled_pass']['mxm_stage_ms_both_streams'],2))"
done; done
