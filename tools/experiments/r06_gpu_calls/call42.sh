#!/bin/bash
# round 6, GPU call 42: LDS counters of the GEMM family (bank conflicts, unaligned stalls, LDS busy) + wait split
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
SHORT="--no-cpu-baseline --no-inference --no-extras"
cd /tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" \
           "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d /tmp/l$i -o r -- python $R/bench.py --steps 2 --warmup 1 $SHORT > /tmp/bl$i.log 2>&1
  echo "== pass $i: $set"
  for k in conv_gemm2_kernel gemm_tn_batch_kernel "gemm_tn_kernel" "conv_gemm_kernel<1" bigru_fwd; do python $R/tools/pmc_generic.py $(find /tmp/l$i -name "*.db" | head -1) "$k" 2>&1 | head -12; done
done > $O/r06_call42.log 2>&1
tail -60 $O/r06_call42.log
