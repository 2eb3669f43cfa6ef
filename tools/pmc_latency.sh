#!/bin/bash
# Average latency of scalar-memory, vector-memory and LDS instructions of the headline bench:
# SQ_INST_LEVEL_* accumulates the number of such instructions in flight per cycle; divided by the number issued it is
# the mean time one of them is outstanding.  Usage on the GPU box: tools/pmc_latency.sh <tag> -> gpurun_out/lat_<tag>/
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/lat_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
  echo "pass $name ok"
}
pass smem SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES && \
pass vmem SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES && \
pass lds SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
    last = sorted({int(r["Dispatch_Id"]) for r in rows})[-2:]
    tot = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in last: tot[r["Counter_Name"]] += float(r["Counter_Value"]) / len(last)
    print(f.split("/")[-3], {k: round(v) for k, v in tot.items()})
PY
