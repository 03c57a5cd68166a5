#!/bin/bash
# round 6, GPU call 22: DYNAMIC instruction mix and wait split of the decoder kernels (SQ counters), to decide what a diet can buy
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
SHORT="--no-cpu-baseline --no-inference --no-extras"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/r06_sq_counters.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d /tmp/q$i -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/bq$i.log 2>&1
  echo "== pass $i: $set"
  python $R/tools/pmc_generic.py $(find /tmp/q$i -name "*.db" | head -1) decoder3 2>&1 | tail -30
done > $O/r06_call22.log 2>&1
cat $O/r06_call22.log | tail -120
