#!/bin/bash
# A/B of the pre-pass that runs ahead (VOLLY_PREPASS_AHEAD=0: in place on the caller's stream; default: ahead while the
# previous launch is still running): wall-clock Mrays/s and kernel ms of bench.py's workloads, alternating.
#   usage (repo root, GPU box): bash profiles/r03_ahead_ab.sh <out dir under gpurun_out>
set -u
TAG=${1:-r03_ahead}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {   # name, mode, bench args...
  local name=$1 mode=$2; shift 2
  timeout -k 10 200 env VOLLY_PREPASS_AHEAD=$mode python -u bench.py --steps 20 --warmup 5 --no-device-prepare --no-cpu-baseline --no-end-to-end "$@" > $OUT/${name}_$mode.json 2> $OUT/${name}_$mode.err || { echo "$name $mode failed"; tail -5 $OUT/${name}_$mode.err; return 1; }
  python - "$OUT/${name}_$mode.json" "$name" "$mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:16s} ahead={sys.argv[3]}  {d['value']:10.1f} Mrays/s  step {d['ms_per_step']:7.4f} ms  kernel {r['kernel_avg_ms']:7.4f} ms")
PY
}
for rep in 1 2; do
for mode in 0 2; do
  run c0 $mode || exit 1
  run c0_naive $mode --mode naive || exit 1
  run c0_startup $mode --camera startup || exit 1
  run c1 $mode --config 1 || exit 1
  run c2 $mode --config 2 || exit 1
  run c3 $mode --config 3 || exit 1
  run c5 $mode --config 5 || exit 1
done
done
