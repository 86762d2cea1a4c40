#!/bin/bash
# Round-3 bench lines on the GPU box: every BASELINE config, the headline's camera regimes, both modes where they differ.
#   usage (repo root): bash profiles/r03_sweep.sh <out dir under gpurun_out>
set -u
TAG=${1:-r03_sweep}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 280 python -u bench.py --steps 20 --warmup 5 --no-device-prepare "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -5 $OUT/bench_$name.err; return 1; }
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
print(f"{sys.argv[2]:18s} {d['value']:10.1f} Mrays/s  kernel {r['kernel_avg_ms']:7.3f} ms  valu_frac {r['frac']:.3f}  hit {c['hit_fraction']:.3f}  steps {c['mean_steps_lo']:.1f}+{c['mean_steps_hi']:.2f}  cpu {d.get('cpu_baseline', {}).get('value')}")
PY
}
run c0 || exit 1
run c0_naive --mode naive --no-cpu-baseline --no-end-to-end || exit 1
for cam in startup inside fill; do
  run c0_$cam --camera $cam || exit 1
  run c0_${cam}_naive --camera $cam --mode naive --no-cpu-baseline --no-end-to-end || exit 1
done
run c1 --config 1 || exit 1
run c2 --config 2 || exit 1
run c2_naive --config 2 --mode naive --no-cpu-baseline --no-end-to-end || exit 1
run c3 --config 3 --no-end-to-end || exit 1
run c3_naive --config 3 --mode naive --no-end-to-end || exit 1
run c3_v16 --config 3 --views 16 --no-cpu-baseline --no-end-to-end || exit 1
run c3_p64 --config 3 --noise-log2-inv-p 6 --no-end-to-end || exit 1
run c3_p1024 --config 3 --noise-log2-inv-p 10 --no-end-to-end || exit 1
run c4 --config 4 || exit 1
run c5 --config 5 || exit 1
