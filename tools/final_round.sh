#!/bin/bash
# What a round leaves under profiles/ besides the counter rounds (tools/profile_all.sh), run after their summaries have
# been made (tools/summarize_profile.py), so that every bench line replays its counters: the bench line of every
# configuration, the results table, the partition scaling and the GPU test log.
# Usage on the GPU box: tools/final_round.sh   -> gpurun_out/final/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
S=scenes
timeout -k 10 300 python bench.py --cpu-whole-frame > $OUT/chess_bench.json 2> $OUT/chess_bench.err || exit 1
for s in fractal cone-fractal spotlight house spheres; do
  timeout -k 10 200 python bench.py --scene $S/$s.gml --no-cpu-baseline --no-host-path > $OUT/${s}_bench.json 2> $OUT/${s}_bench.err || exit 1
done
timeout -k 10 200 python bench.py --scene $S/large.gml --width 7680 --height 4320 --depth 8 --no-cpu-baseline --no-host-path > $OUT/large_bench.json 2> $OUT/large_bench.err || exit 1
echo "bench lines done"
timeout -k 10 300 python tools/part_scaling.py > $OUT/part_scaling.txt 2> $OUT/part_scaling.err || exit 1
echo "partition scaling done"
timeout -k 10 600 python tools/results_table.py > $OUT/results_table.md 2> $OUT/results_table.err || exit 1
echo "results table done"
timeout -k 10 900 python -u -m pytest tests -v --timeout 300 --timeout-method thread -m gpu > $OUT/gpu_tests.log 2>&1 || exit 1
tail -1 $OUT/gpu_tests.log
