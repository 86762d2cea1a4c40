#!/bin/bash
# The bench lines of DESIGN.md section 5a (profiles/make_tables.py reads them): one default-flag run of
# bench.py per workload, on the GPU box.   usage (repo root): bash profiles/collect_bench_lines.sh <round tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bench_$TAG
mkdir -p $OUT
run() { key=$1; shift; timeout -k 10 400 python3 $ROOT/bench.py "$@" 2> $OUT/bench_$key.err | grep '^{' > $OUT/bench_$key.json || { echo "$key failed"; tail -3 $OUT/bench_$key.err; exit 1; }; echo "$key ok"; }
run c0 --config 0
run c0_naive --config 0 --mode naive
run c1 --config 1
run c2 --config 2
run c2_naive --config 2 --mode naive
run c3 --config 3
run c3_p64 --config 3 --noise-log2-inv-p 6
run c3_p1024 --config 3 --noise-log2-inv-p 10
run c4 --config 4
run c5 --config 5
