#!/bin/bash
# Tuning sweep of the bound-emission thresholds (environment overrides read by the flattener).
# Usage on the GPU box: tools/sweep_bounds.sh [scene]  -> gpurun_out/sweep_bounds.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SCENE=${1:-scenes/chess.gml}
: > gpurun_out/sweep_bounds.txt
for op in ${OPS:-0.3 0.5 0.7 0.9}; do
  for lf in ${LEAVES:-0.1 0.25 0.5 0.8}; do
    for bias in 0.8; do
      GMLRT_BOUND_OP_RATIO=$op GMLRT_BOUND_LEAF_RATIO=$lf GMLRT_BOX_BIAS=$bias timeout -k 10 120 python bench.py --scene $SCENE --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_one.json 2> gpurun_out/sweep_one.err || { echo "op $op leaf $lf FAILED" >> gpurun_out/sweep_bounds.txt; exit 1; }
      python - "$op" "$lf" "$bias" <<'PY' >> gpurun_out/sweep_bounds.txt
import json,sys
d=json.load(open("gpurun_out/sweep_one.json"))
print("op", sys.argv[1], "leaf", sys.argv[2], "bias", sys.argv[3], "Mrays/s", d["value"], "ms", d["ms_per_step"])
PY
    done
  done
done
cat gpurun_out/sweep_bounds.txt
