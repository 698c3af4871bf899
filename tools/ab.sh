#!/bin/bash
# A/B two builds of libzigp.so on the SAME GPU box (devices differ by several %): usage tools/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do for L in $A $B; do
  ZIGP_LIB=$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['roofline']['per_kernel_tflops'].items()})"
done; done
