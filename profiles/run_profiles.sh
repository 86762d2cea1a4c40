#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel on the GPU box.
#   usage (from the repo root, on the GPU box):  bash profiles/run_profiles.sh <tag> [bench args...]
# Each counter group is its own rocprofv3 run (kernel-trace/stats never combined with --pmc).
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-device-prepare $*"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace.log; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc run $grp failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
done
echo done; find $OUT -name '*.csv' | head -30
