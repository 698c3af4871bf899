#!/bin/bash
# Evidence of a round, collected on ONE GPU box (from the repo root): tools/collect_round.sh <tag> [dense] [kron]
#   dense: the full bench line, a rocprofv3 --kernel-trace --stats run of the same command (overlap off), the HBM-traffic and MFMA-busy
#          PMC passes (separate passes, --pmc with --kernel-trace only), the LDS counters
#   kron:  kernel stats of the Kronecker configurations (tools/kron_prof.py)
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=$1; shift
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
stats() {   # stats <out prefix> <program args...>: kernel stats CSV of one traced run
  local out=$1; shift
  rm -rf /tmp/prof_$$ && (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -- "$@" > $ROOT/gpurun_out/${out}.log 2>&1)
  local f=$(find /tmp/prof_$$ -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $ROOT/gpurun_out/${out}_kernel_stats.csv
}
for what in "$@"; do
  if [ "$what" = dense ]; then
    timeout -k 10 900 python3 bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || echo "bench failed"
    stats ${TAG}_bench python3 $ROOT/bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-other-configs --no-overlap --profile-steps 1
    grep '^{' gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_under_rocprof.json
    bash tools/pmc_run.sh ${TAG} || echo "pmc_run failed"
    bash tools/pmc_mfma.sh ${TAG} || echo "pmc_mfma failed"
    bash tools/pmc_lds.sh > gpurun_out/${TAG}_pmc_lds.log 2>&1 || echo "pmc_lds failed"
  fi
  if [ "$what" = kron ]; then
    for m in full_res mb mb10x100; do
      stats ${TAG}_kron_${m} python3 $ROOT/tools/kron_prof.py $m 50
    done
  fi
done
ls -la gpurun_out | grep ${TAG}_
