"""CPU, world_size 2 over gloo: the data-parallel exchange (tacotron_amd/dist.py) reproduces the 2N-batch gradient and
keeps replicas identical.  The model arithmetic in this test is the CPU restatement (test infrastructure); the code under
test is the reducer + the SUM (not mean) semantics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import taco_numpy as on
from oracle import taco_torch as ot
from tests.util import small_case


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from tacotron_amd.dist import GradReducer, init_from_env
    r, w, _ = init_from_env('gloo')
    assert (r, w) == (rank, world)
    V, rr, B, Tt, Td = 15, 2, 4, 8, 4
    p = on.init_params(V, rr, seed=3, perturb=0.2)
    inp, masks = small_case(r=rr, V=V, B=B, Tt=Tt, Td=Td, seed=2)
    sl = slice(rank * 2, rank * 2 + 2)
    my_inp = {k: v[sl].astype(np.float64) if v.dtype == np.float32 else v[sl] for k, v in inp.items()}
    my_masks = {k: (v[:, sl] if k == 'sample' else v[sl]).astype(np.float64) for k, v in masks.items()}
    loss, _, _, _, grads = ot.loss_and_grads(p, my_inp, rr, Td, my_masks)
    flat = torch.tensor(np.concatenate([grads[n].reshape(-1) for n, _, _ in on.param_spec(V, rr)]), dtype=torch.float32)
    lt = torch.tensor([loss, 0.0, 0.0], dtype=torch.float32)
    red = GradReducer(bucket_floats=1 << 20)      # several buckets for 6.9 M parameters
    assert red.world == 2
    red.all_reduce(flat, lt)
    # replicated clip + Adam on the reduced gradient
    params = torch.tensor(on.flatten_params(p, V, rr), dtype=torch.float32)
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    ot.clip_adam_step({'w': params}, {'w': flat}, {'w': m}, {'w': v}, 1, 5e-4)
    q.put((rank, flat.numpy(), float(lt[0]), params.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sum_allreduce_matches_full_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    # reference: one process, full batch of 4
    V, rr, B, Tt, Td = 15, 2, 4, 8, 4
    p = on.init_params(V, rr, seed=3, perturb=0.2)
    inp, masks = small_case(r=rr, V=V, B=B, Tt=Tt, Td=Td, seed=2)
    inp64 = {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in inp.items()}
    loss, _, _, _, grads = ot.loss_and_grads(p, inp64, rr, Td, {k: v.astype(np.float64) for k, v in masks.items()})
    full = np.concatenate([grads[n].reshape(-1) for n, _, _ in on.param_spec(V, rr)])
    for rank, flat, l, params in res:
        assert np.linalg.norm(flat - full) / np.linalg.norm(full) < 1e-6     # SUM of per-rank grads == 2N-batch grad
        assert abs(l - loss) < 1e-5 * loss
    assert np.array_equal(res[0][1], res[1][1])          # identical reduced gradients on both ranks
    assert np.array_equal(res[0][3], res[1][3])          # replicas stay bit-identical after clip + Adam


def _flag_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from tacotron_amd.dist import GradReducer, init_from_env
    init_from_env('gloo')
    red = GradReducer(bucket_floats=1 << 10)
    g = torch.ones(3000)
    err = torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32)   # rank 1's forward kernel timed out; the word is sticky
    seen = []
    for _ in range(40):                                                # > 32 steps: a SUM would wrap 2^32 -> 0 at world 2
        loss = torch.tensor([1.0, 0.0, 0.0])
        red.all_reduce(g.clone(), loss, err)
        seen.append(err.tolist())
    q.put((rank, seen, float(loss[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_error_words_stay_set_across_many_reduced_steps():
    """the sticky decoder error words ride along with MAX, so a flag raised on ONE rank is seen by every
    rank as exactly 1 for as long as nobody clears it (SUM multiplied it by the world size per step and wrapped to 0)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    for rank, seen, loss in res:
        assert all(s == [1, 0] for s in seen), (rank, seen[:3], seen[-3:])
        assert loss == 2.0                                              # the loss is still a SUM
