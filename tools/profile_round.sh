#!/bin/bash
# The evidence set for profiles/: kernel-trace stats of a bench command, then counter passes
# (one rocprofv3 run each; never more than 8 SQ, 4 TCC or 2 GRBM counters a pass).
# Usage on the GPU box: tools/profile_round.sh <tag> [bench.py arguments that name the workload, e.g.
#   --scene scenes/fractal.gml --depth 3]                                  -> gpurun_out/prof_<tag>/
TAG=${1:-run}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-path "$@" > $OUT/bench_under_trace.json 2> $OUT/trace.err || { echo "trace failed"; tail -3 $OUT/trace.err; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path $WORKLOAD > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
  echo "pass $name ok"
}
WORKLOAD="$*"
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM && \
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVES && \
pass mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU && \
pass tcc_rd FETCH_SIZE && \
pass tcc_wr WRITE_SIZE && \
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
if [ -n "$FULL_BENCH" ]; then EXTRA="--cpu-whole-frame"; else EXTRA="--no-cpu-baseline --no-host-path"; fi
cd $ROOT && timeout -k 10 300 python bench.py $WORKLOAD $EXTRA > $OUT/bench_default.json 2> $OUT/bench_default.err
ls $OUT
