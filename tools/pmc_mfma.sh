#!/bin/bash
# MFMA pipe utilisation per kernel: separate rocprofv3 --pmc passes (no trace domains besides --kernel-trace) of a short cfg3 run.
# usage (GPU box, repo root): tools/pmc_mfma.sh <tag>  ->  gpurun_out/<tag>_pmc_mfma_util.json
TAG=${1:-pmc}
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/$TAG/$C -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu-baseline --no-other-configs --profile-steps 0 --no-overlap --rows ${ROWS:-262144} > $ROOT/gpurun_out/$TAG/$C.log 2>&1 || exit 1
done
cd $ROOT
python3 tools/pmc_mfma.py gpurun_out/$TAG > gpurun_out/${TAG}_pmc_mfma_util.json
