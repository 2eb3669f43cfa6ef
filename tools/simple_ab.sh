#!/bin/bash
# Kernel throughput of a cheap, a medium and the headline scene for each variant library named on the command line
# (rust-raytracer_amd/variants/libgmlrt_<name>.so).  Usage on the GPU box: tools/simple_ab.sh a b ...
cd "$(dirname "$0")/.."
for v in "$@"; do for s in ${SCENES:-large spotlight chess}; do
  GMLRT_LIB=$PWD/rust-raytracer_amd/variants/libgmlrt_$v.so timeout -k 10 100 python bench.py --scene scenes/$s.gml --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$s', d['value'], d['ms_per_step'])"
done; done
