#!/bin/bash
# Memory-side traffic of the render kernel (one counter per rocprofv3 run).
# Usage on the GPU box: tools/pmc_mem.sh <tag>  -> gpurun_out/pmc_<tag>/{tcc_rd,tcc_wr}
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
  grep -h render_kernel $OUT/$name/*/*_counter_collection.csv | awk -F, -v n=$name '{s+=$(NF-2); c++} END {printf "%s: %.1f MB per launch (%d launches)\n", n, s/c/1024, c}'
}
pass tcc_rd FETCH_SIZE && pass tcc_wr WRITE_SIZE
