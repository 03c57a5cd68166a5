#!/bin/bash
# round 5, GPU call 2: (a) weight-gradient kernels on bf16x3: parity + tile / split sweep; (b) where the bf16x3 NN k-loop loses its time (lab builds)
mkdir -p gpurun_out
{
echo "== ops (tn + gemm)"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -5
echo "== lab: NN kernel 16x4, plain shapes, steady state"
for l in hip lab_nosplit lab_nomfma lab_nodma lab_nobar lab_nosplitnodma; do echo "-- lib $l"; TACO_LIB=$PWD/tacotron_amd/libtaco_$l.so timeout 120 python tools/gemm_variants.py 16x4 2>&1 | grep -v amdgpu.ids; done
echo "-- lib hip, fp32 form"; TACO_GEMM2_BF16X=0 timeout 120 python tools/gemm_variants.py 16x4 32x2 2>&1 | grep -v amdgpu.ids
echo "== family trace, TN sweeps"
for cfg in "BM=0 BLOCKS=3072" "BM=128 BLOCKS=3072" "BM=128 BLOCKS=1536" "BM=128 BLOCKS=768" "BM=64 BLOCKS=1536"; do
  eval $cfg; echo "-- TACO_TN_BM=$BM TACO_TN_BLOCKS=$BLOCKS"
  TACO_TN_BM=$BM TACO_TN_BLOCKS=$BLOCKS timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| tn"
done
echo "-- fp32 TN for reference (BF16X=0)"; TACO_GEMM2_BF16X=0 timeout 300 python tools/family_trace.py 2>&1 | grep -E "^step|^sum| tn"
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -5
} > gpurun_out/r05_call2.log 2>&1
tail -c 3000 gpurun_out/r05_call2.log
