#!/bin/bash
# The check build of the look before the leaf test (-DGMLRT_DIAG -DGMLRT_LEAF_LOOK_CHECK, tools/build_variant.sh) on every
# configuration scene: per primitive type, how many leaf tests the look would skip and how many lanes with an Entry it
# turned away (must be 0).  Usage on the GPU box: tools/look_check.sh <variant name>  -> gpurun_out/look_check.txt
cd "$(dirname "$0")/.."
V=${1:-lookchk}
: > gpurun_out/look_check.txt
for s in chess fractal cone-fractal spotlight house spheres large; do
  echo "== $s" >> gpurun_out/look_check.txt
  GMLRT_LIB=$PWD/rust-raytracer_amd/variants/libgmlrt_$V.so GMLRT_VERBOSE=1 timeout -k 10 200 python tools/levels_trace.py $s 1 2>&1 | grep -E "leaf tests|kernel_ms|no lane has" >> gpurun_out/look_check.txt || exit 1
done
cat gpurun_out/look_check.txt
