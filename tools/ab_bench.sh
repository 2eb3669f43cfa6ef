#!/bin/bash
# Runs bench.py once per variant library given on the command line (names under rust-raytracer_amd/variants/).
# Usage on the GPU box:  tools/ab_bench.sh w1 w2 ...   -> gpurun_out/ab_<name>.json
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  GMLRT_LIB=$PWD/rust-raytracer_amd/variants/libgmlrt_$v.so timeout -k 10 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.load(open(f"gpurun_out/ab_{v}.json"))
print(v, "Mrays/s", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
done
