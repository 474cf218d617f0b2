#!/bin/bash
# Build an instrumented variant of the device library next to the product one:  tools/build_variant.sh <name> <flags...>
# -> allegro_amd/liballegro_amd_<name>.so (git-ignored; select it with ALLEGRO_AMD_LIBRARY=<path>)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC="aa_gemm.hip aa_tp.hip aa_tp_spec.hip aa_tp_op.hip aa_tp_dense.hip aa_edge.hip aa_fused.hip aa_model.hip aa_nl.hip aa_hostfile.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -I include -I allegro_amd/csrc "$@" \
  $(for f in $SRC; do echo allegro_amd/csrc/$f; done) -o allegro_amd/liballegro_amd_$NAME.so
echo built allegro_amd/liballegro_amd_$NAME.so
