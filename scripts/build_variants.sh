#!/bin/bash
# Build A/B variants of libsvo_amd.so into octree-raymarcher_amd/build/ (experiments only).
# usage: scripts/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"..."
set -e
cd "$(dirname "$0")/../octree-raymarcher_amd"
mkdir -p build
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize $defs -shared -o build/libsvo_$name.so csrc/world.cpp csrc/terrain.cpp csrc/device.hip csrc/builder.hip csrc/shade.hip -lpthread &
done
wait
ls -la build/*.so
