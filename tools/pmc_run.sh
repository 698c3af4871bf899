#!/bin/bash
# HBM traffic per kernel launch: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE separately, no trace domains), on a short
# run of the cfg3 step.  usage (on the GPU box, from the repo root): tools/pmc_run.sh <tag>  ->  gpurun_out/<tag>_pmc_hbm_traffic.json
TAG=${1:-pmc}
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/$C -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu-baseline --no-other-configs --profile-steps 0 --no-overlap --rows ${ROWS:-262144} > $ROOT/gpurun_out/$TAG/$C.log 2>&1 || exit 1
done
cd $ROOT
F=$(find gpurun_out/$TAG/FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/$TAG/WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $F $W > gpurun_out/${TAG}_pmc_hbm_traffic.json
