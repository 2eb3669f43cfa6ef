#!/bin/bash
# Builds a variant of libgmlrt.so with extra kernel compile flags, for A/B runs on the GPU box:
#   tools/build_variant.sh w4 "-DGMLRT_WAVES_PER_SIMD=4"   ->  rust-raytracer_amd/variants/libgmlrt_w4.so
# Select it at run time with GMLRT_LIB=<path>.
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rust-raytracer_amd/csrc
mkdir -p $ROOT/rust-raytracer_amd/variants $C/build
make -s -C $C -j8 ../libgmlrt.so >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wno-unused-parameter $FLAGS \
  -c $C/gmlrt_kernels.hip -o $C/build/kernels_$NAME.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|Scratch|Occupancy|Spill|error" | tr '\n' ' '
echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/rust-raytracer_amd/variants/libgmlrt_$NAME.so \
  $C/build/gml_front.o $C/build/gmlrt_scene.o $C/build/gmlrt_program.o $C/build/gmlrt_api.o $C/build/kernels_$NAME.o -pthread
