set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc2; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-device-prepare"
i=0
for grp in "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD TA_TA_BUSY_sum TA_BUSY_avr" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
echo ok
