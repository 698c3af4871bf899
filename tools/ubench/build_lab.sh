#!/bin/bash
# build tools/ubench/gemm_lab (gfx950) and print the resource usage of its kernels; the assembly goes to /tmp/lab/<name>.s
# usage: build_lab.sh [name=gemm_lab] [-DFLAG ...]
set -e
D=$(cd "$(dirname "$0")" && pwd)
NAME=${1:-gemm_lab}; shift || true
mkdir -p /tmp/lab/$NAME && cd /tmp/lab/$NAME
/opt/rocm/bin/hipcc -O3 "$@" --offload-arch=gfx950 -std=c++17 -I$D/../../zero-inflated-gp_amd/csrc $D/gemm_lab.hip -o /tmp/lab/$NAME/gemm_lab -save-temps=obj 2>&1 | grep -v "argument unused" || true
cp gemm_lab-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/lab/$NAME.s; cp /tmp/lab/$NAME/gemm_lab $D/$NAME
grep "^_Z.*:$\|^_Z.*: *;\|; NumVgprs\|; ScratchSize\|; Occupancy" /tmp/lab/$NAME.s | sed 's/: *;.*//' | paste - - - - | awk '{print $1, $4, $7, $10}'
