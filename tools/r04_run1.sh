# Round-4 GPU run 1: parity suite on the refactored DP / decoder-mode code, labelled family trace, per-shape PMC of the GEMM kernels,
# forced world-1 RCCL run with and without the high-priority communication stream.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/r04a_pytest.txt
python tools/family_trace.py > $O/r04a_family_trace.txt 2>&1
python tools/gemm_pmc.py $O/r04a_gemm_order.json > $O/r04a_gemm_shapes.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d /tmp/g1 -o r -- python $R/tools/gemm_pmc.py /tmp/o1.json > /tmp/g1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/g2 -o r -- python $R/tools/gemm_pmc.py /tmp/o2.json > /tmp/g2.log 2>&1
cd $R
python tools/gemm_pmc_read.py $(find /tmp/g1 -name "*.db" | head -1) /tmp/o1.json > $O/r04a_gemm_pmc_mfma.txt 2>&1
python tools/gemm_pmc_read.py $(find /tmp/g2 -name "*.db" | head -1) /tmp/o2.json > $O/r04a_gemm_pmc_sq.txt 2>&1
tail -5 /tmp/g1.log /tmp/g2.log > $O/r04a_gemm_pmc_logs.txt 2>&1
SHORT="--no-cpu-baseline --no-inference --no-extras"
python bench.py $SHORT > $O/r04a_bench_plain.json 2> $O/r04a_bench_plain.err
TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $SHORT > $O/r04a_bench_dist_hi.json 2> $O/r04a_bench_dist_hi.err
TACO_COMM_PRIORITY=0 TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 $SHORT > $O/r04a_bench_dist_lo.json 2> $O/r04a_bench_dist_lo.err
GPU_MAX_HW_QUEUES=8 TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 $SHORT > $O/r04a_bench_dist_hi_q8.json 2> $O/r04a_bench_dist_hi_q8.err
cat $O/r04a_pytest.txt | tail -3
