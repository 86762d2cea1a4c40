#!/bin/bash
# config 3 with the linear and the 2x2x2-blocked brick layout: kernel time (interleaved A/B) and L2 / TA counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r02_layout; mkdir -p $OUT; export TMPDIR=/tmp
L=$ROOT/volume-renderer_amd/lib
cd $ROOT
for c in 3 1; do
  timeout -k 10 300 python3 profiles/ab_bench.py --config $c --rounds 9 linear=$L/libvolly_gpu.so blocks=$L/variants/libvolly_gpu_blocks.so > $OUT/ab_c$c.log 2>&1 || exit 1
  tail -2 $OUT/ab_c$c.log
done
cd /tmp
for v in linear blocks; do
  lib=$L/libvolly_gpu.so; [ $v = blocks ] && lib=$L/variants/libvolly_gpu_blocks.so
  export VOLLY_GPU_LIB=$lib
  timeout -s KILL 200 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TA_TA_BUSY_sum TA_BUSY_avr SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_$v -o pmc -- python3 $ROOT/bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/pmc_$v.log 2>&1 || { echo "pmc $v failed"; tail -3 $OUT/pmc_$v.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -o pmc -- python3 $ROOT/bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/fetch_$v.log 2>&1 || exit 1
done
echo done
