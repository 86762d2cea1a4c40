#!/bin/bash
# Per-ray fixed cost of the headline launch (16 x 4K, five planes), taken apart with bench.py's own workload:
#   far      orbit radius 60: the volume is a dot, nearly every tile is sky -> workgroup dispatch + the sky blocks' stores
#   far_full the same views with the sky cull off: full per-ray set-up, 0 steps, stores
#   startup  the reference's start-up camera: set-up, 1.2 steps per ray, stores
#   usage (repo root, GPU box): bash profiles/r03_fixed_cost.sh <out dir under gpurun_out>
set -u
TAG=${1:-r03_fixed}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {   # name, env assignment or "-", bench args...
  local name=$1 envs=$2; shift 2
  if [ "$envs" = "-" ]; then
    timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-device-prepare --no-cpu-baseline --no-end-to-end "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }
  else
    timeout -k 10 200 env $envs python -u bench.py --steps 20 --warmup 5 --no-device-prepare --no-cpu-baseline --no-end-to-end "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -5 $OUT/$name.err; return 1; }
  fi
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
print(f"{sys.argv[2]:22s} kernel {r['kernel_avg_ms']:7.3f} ms  {d['value']:10.1f} Mrays/s  hit {c['hit_fraction']:.3f}  steps {c['mean_steps_lo']:.2f}+{c['mean_steps_hi']:.2f}")
PY
}
for mode in beam naive; do
  run far_$mode - --radius 60 --mode $mode || exit 1
  run far_full_$mode VOLLY_SKY_CULL=0 --radius 60 --mode $mode || exit 1
  run startup_$mode - --camera startup --mode $mode || exit 1
done
