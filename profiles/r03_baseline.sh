#!/bin/bash
# Round-3 baseline on the GPU box: the suite, the bench line of every camera regime, counters of the pure octree traversal.
#   usage (repo root): bash profiles/r03_baseline.sh <out tag>
set -u
TAG=${1:-r03_base}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for cam in orbit startup inside fill; do
  for mode in beam naive; do
    extra=""
    [ "$mode" = naive ] && extra="--no-cpu-baseline --no-end-to-end"
    timeout -k 10 280 python -u bench.py --camera $cam --mode $mode --steps 10 --warmup 4 --no-device-prepare $extra > $OUT/bench_c0_${cam}_${mode}.json 2> $OUT/bench_c0_${cam}_${mode}.err || { echo "bench $cam $mode failed"; tail -5 $OUT/bench_c0_${cam}_${mode}.err; exit 1; }
    echo "bench $cam $mode: $(python -c "import json,sys; d=json.load(open('$OUT/bench_c0_${cam}_${mode}.json')); print(d['value'], d['roofline']['frac'], d['config']['hit_fraction'], d['config']['mean_steps_lo'], d['config']['mean_steps_hi'], d['roofline']['kernel_avg_ms'])")"
  done
done
