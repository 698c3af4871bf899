#!/bin/bash
# same-box A/B of environment settings: tools/ab_envs.sh ROUNDS "A=1 B=2" "A=0" ...   (each argument one setting, "-" = none)
R=$1; shift
for r in $(seq 1 $R); do for V in "$@"; do
  E="$V"; [ "$V" = "-" ] && E=""
  echo -n "[$V]  "; env $E timeout -k 10 300 python tools/ab_cfgs.py 2>&1 | tail -1
done; done
