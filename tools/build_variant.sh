#!/bin/bash
# Build a variant of libzigp.so next to the product library for a same-box A/B (tools/ab.sh): tools/build_variant.sh NAME [-DFLAG ...]
# -> zero-inflated-gp_amd/lib/libzigp_NAME.so  (git-ignored; travels to the GPU box with the snapshot; select with ZIGP_LIB=...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 "$@" --offload-arch=gfx950 -std=c++17 -shared -fPIC \
  -o "$ROOT/zero-inflated-gp_amd/lib/libzigp_$NAME.so" "$ROOT/zero-inflated-gp_amd/csrc/zigp_lib.hip"
echo "$ROOT/zero-inflated-gp_amd/lib/libzigp_$NAME.so"
