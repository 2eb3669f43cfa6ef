#!/bin/bash
# Registers, scratch, LDS and occupancy of the render kernels as the compiler reports them (no GPU needed):
#   tools/kernel_resources.sh > profiles/<round>_kernel_resources.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rust-raytracer_amd/csrc
echo "# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage gmlrt_kernels.hip"
echo "# kernel sources sha256/16: $(python3 $ROOT/tools/source_hash.py)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wno-unused-parameter --cuda-device-only \
  -c $C/gmlrt_kernels.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "remark:" | sed -E 's/^.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//'
echo "# LDS is dynamic (extern __shared__): operand-stack lengths (max_stack x 512 B) + 18 x 256 x 8 B of shading context per 256-thread block"
