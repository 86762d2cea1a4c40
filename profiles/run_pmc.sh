#!/bin/bash
# rocprofv3 evidence for one bench.py workload on the GPU box: a kernel-trace/stats run, then one
# run per counter group (never combined: gpurun refuses --pmc together with tracing domains, and a
# pass holds at most 8 SQ_ / 4 TCC_ / 2 TA_ / 2 GRBM_ counters).
#   usage (repo root, on the GPU box):  bash profiles/run_pmc.sh <round tag> <workload key> [bench args...]
#   e.g.  bash profiles/run_pmc.sh r02 config0_beam --config 0
# Output: gpurun_out/prof_<tag>/<key>/...; condense with  python profiles/summarize.py <tag>
set -u
TAG=$1; KEY=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG/$KEY
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-device-prepare $*"
echo "$ARGS" > $OUT/bench_args.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "$KEY: trace run failed"; tail -5 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log > $OUT/bench_line.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_WRITE_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR TA_TA_BUSY_sum TA_BUSY_avr" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$i.log 2>&1 || { echo "$KEY: pmc pass $i ($grp) failed"; tail -5 $OUT/pmc_$i.log; exit 1; }
  echo "$KEY: pass $i ok"
done
echo "$KEY: done"
