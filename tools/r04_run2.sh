# Round-4 GPU run 2: the new parity tests, HW-queue effect on the plain step, forced world-1 RCCL variants
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_dist.py "tests/test_gpu_model.py::test_error_words_are_sticky_and_guard_the_update" -m gpu -x -q -s 2>&1 | tail -150) > $O/r04b_pytest.txt
SHORT="--no-cpu-baseline --no-inference --no-extras"
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python bench.py $SHORT > $O/r04b_bench_plain_q$q.json 2> $O/r04b_bench_plain_q$q.err
done
GPU_MAX_HW_QUEUES=8 python bench.py $SHORT > $O/r04b_bench_plain_q8b.json 2>/dev/null
GPU_MAX_HW_QUEUES=4 python bench.py $SHORT > $O/r04b_bench_plain_q4b.json 2>/dev/null
TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $SHORT > $O/r04b_bench_dist.json 2> $O/r04b_bench_dist.err
TACO_COMM_PRIORITY=0 TACO_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 $SHORT > $O/r04b_bench_dist_lo.json 2> $O/r04b_bench_dist_lo.err
tail -5 $O/r04b_pytest.txt
