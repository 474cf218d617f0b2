#!/bin/bash
# Build a variant of the device library next to the product one:  tools/build_variant.sh <name> <defines...>
# -> allegro_amd/liballegro_amd_<name>.so (git-ignored; select it with ALLEGRO_AMD_LIBRARY=<path>; `tools/gpu_round.sh ab` alternates
# liballegro_amd_old.so with the product library on one box).  Uses the package's own parallel, content-hashed build.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
cp allegro_amd/liballegro_amd.so /tmp/liballegro_amd_product.so
AA_BUILD_DEFINES="$*" python -m allegro_amd.build --force | grep -E 'error|built' || true
mv allegro_amd/liballegro_amd.so allegro_amd/liballegro_amd_$NAME.so
mv /tmp/liballegro_amd_product.so allegro_amd/liballegro_amd.so
touch allegro_amd/liballegro_amd.so allegro_amd/liballegro_amd_torch.so
echo built allegro_amd/liballegro_amd_$NAME.so
