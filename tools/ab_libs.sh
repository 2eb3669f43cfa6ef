#!/bin/bash
# A/B of library builds on the headline frame, alternating, ROUNDS times: tools/ab_libs.sh <name> <name> ...
# (a name is rust-raytracer_amd/variants/libgmlrt_<name>.so; "head" is the library in the tree)
cd "$(dirname "$0")/.."
for i in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    if [ $v = head ]; then lib=$PWD/rust-raytracer_amd/libgmlrt.so; else lib=$PWD/rust-raytracer_amd/variants/libgmlrt_$v.so; fi
    GMLRT_LIB=$lib timeout -k 10 100 python bench.py ${SCENE:+--scene scenes/$SCENE.gml} --steps 10 --warmup 2 --no-cpu-baseline --no-host-path 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['kernel_ms'])" || exit 1
  done
done
