#!/bin/bash
# round 6, GPU call 23: mat-vec FMAs of the persistent decoder as v_pk_fma_f32 on row pairs (libtaco_pk.so) vs the product build, same box
mkdir -p gpurun_out
{
echo "== parity on the pk build (decoder tests)"
TACO_LIB=$PWD/tacotron_amd/libtaco_pk.so timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden or full_size or geometries or peaked or beyond or infer" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_run.sh hip pk
} > gpurun_out/r06_call23.log 2>&1
cat gpurun_out/r06_call23.log
