#!/bin/bash
# same-box A/B of values of one environment variable: tools/ab_env.sh ROUNDS VAR v1 v2 ...
R=$1; VAR=$2; shift 2
for r in $(seq 1 $R); do for V in "$@"; do
  echo -n "$VAR=$V  "; env $VAR=$V timeout -k 10 300 python tools/ab_cfgs.py 2>&1 | tail -1
done; done
