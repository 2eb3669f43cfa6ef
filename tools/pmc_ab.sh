#!/bin/bash
# One counter pass (instruction counts and busy cycles) of the headline frame, or of SCENE_ARGS, for each variant library
# named on the command line ("main" = the in-tree library).  Prints the means over the two timed launches.
# Usage on the GPU box: tools/pmc_ab.sh main nolook ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then export GMLRT_LIB=$ROOT/rust-raytracer_amd/libgmlrt.so; else export GMLRT_LIB=$ROOT/rust-raytracer_amd/variants/libgmlrt_$v.so; fi
  OUT=$ROOT/gpurun_out/pmcab_$v; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path $SCENE_ARGS > $OUT/bench.json 2> $OUT/err.txt || { echo "$v failed"; tail -3 $OUT/err.txt; exit 1; }
  python3 - "$v" $OUT <<'PY'
import csv, glob, sys, collections, json
v, out = sys.argv[1], sys.argv[2]
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(out + "/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
    timed = sorted({int(r["Dispatch_Id"]) for r in rows})[-2:]
    for r in rows:
        if int(r["Dispatch_Id"]) in timed:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
c = {k: tot[k] / cnt[k] for k in tot}
ms = json.load(open(out + "/bench.json"))["roofline"]["kernel_ms"]
print("%-10s kernel %.3f ms | VALU %.0f M  SALU %.0f M  SMEM %.0f M | lanes %.1f | VALU-active quad-cycles %.0f M, per instr %.2f cycles | wait-inst/wave-cycles %.3f" % (
    v, ms, c["SQ_INSTS_VALU"] / 1e6, c["SQ_INSTS_SALU"] / 1e6, c["SQ_INSTS_SMEM"] / 1e6, c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"],
    c["SQ_ACTIVE_INST_VALU"] / 1e6, 4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
done
