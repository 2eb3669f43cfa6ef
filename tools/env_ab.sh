#!/bin/bash
# A/B of one environment switch on the headline frame (or SCENE=...): tools/env_ab.sh VAR val1 val2 ...
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py ${SCENE:+--scene scenes/$SCENE.gml} --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'])" || exit 1
done
