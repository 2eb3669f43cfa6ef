#!/bin/bash
# Counter rounds (tools/profile_round.sh) for every configuration of BASELINE.json, one after the other; stops at the
# first round that fails.  Usage on the GPU box: tools/profile_all.sh <prefix>   -> gpurun_out/prof_<prefix>_<scene>/
P=${1:-r03}
T=$(cd "$(dirname "$0")" && pwd)
S=$(cd "$T/../scenes" && pwd)
FULL_BENCH=1 $T/profile_round.sh ${P}_chess && \
$T/profile_round.sh ${P}_fractal --scene $S/fractal.gml && \
$T/profile_round.sh ${P}_cone-fractal --scene $S/cone-fractal.gml && \
$T/profile_round.sh ${P}_spotlight --scene $S/spotlight.gml && \
$T/profile_round.sh ${P}_house --scene $S/house.gml && \
$T/profile_round.sh ${P}_spheres --scene $S/spheres.gml && \
$T/profile_round.sh ${P}_large --scene $S/large.gml --width 7680 --height 4320 --depth 8
