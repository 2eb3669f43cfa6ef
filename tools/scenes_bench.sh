#!/bin/bash
# Throughput of every fixture scene at 3840x2160 (depth as the scene file says unless DEPTH is set).
# Usage on the GPU box: tools/scenes_bench.sh  -> gpurun_out/scenes_bench.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/scenes_bench.txt
for s in spheres chess fractal cone-fractal spotlight house large; do
  timeout -k 10 200 python bench.py --scene scenes/$s.gml --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${DEPTH:+--depth $DEPTH} > gpurun_out/scene_one.json 2> gpurun_out/scene_one.err || { echo "$s FAILED" >> gpurun_out/scenes_bench.txt; tail -3 gpurun_out/scene_one.err >> gpurun_out/scenes_bench.txt; continue; }
  python - "$s" <<'PY' >> gpurun_out/scenes_bench.txt
import json,sys
d=json.load(open("gpurun_out/scene_one.json"))
print(f"{sys.argv[1]:14s} {d['value']:10.1f} Mrays/s {d['ms_per_step']:9.3f} ms  rays/frame {d['config']['rays_per_frame']}  leaves {d['config']['leaves']}")
PY
done
cat gpurun_out/scenes_bench.txt
