"""GPU, world_size 2: two processes share GPU 0 and exchange DEVICE tensors over gloo (RCCL refuses two ranks on one
device; the 8-GPU RCCL run is the driver's).  This drives the PRODUCT path -- `Tacotron.step()` of libtaco_hip.so with a
`GradReducer` -- not oracle gradients: per-segment events from `taco_backward`, the all-reduce enqueued on a communication
stream under the rest of the backward pass, loss + decoder error words riding along, guarded clip + Adam.

Checked: after 2 steps on half batches the replicas are bit-identical, and equal (<= 1e-6 rel-L2) to ONE process stepping on
the full batch with the same masks (loss is a SUM, so SUM all-reduce == big-batch gradient)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, TT, TD, R, V = 8, 24, 10, 2, 30
STEPS = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from tacotron_amd.data import synthetic_batch
    batch = synthetic_batch(B, TT, TD, R, V, seed=5, min_len=10)
    rng = np.random.default_rng(99)
    masks = [{'enc_keep1': rng.integers(0, 2, (B, TT, 256)).astype(np.uint8),
              'enc_keep2': rng.integers(0, 2, (B, TT, 128)).astype(np.uint8),
              'dec_keep1': rng.integers(0, 2, (B, TD, 256)).astype(np.uint8),
              'dec_keep2': rng.integers(0, 2, (B, TD, 128)).astype(np.uint8),
              'sample': rng.integers(0, 2, (TD, B)).astype(np.uint8)} for _ in range(STEPS)]
    return batch, masks


def _slice(batch, masks, sl):
    b = {k: v[sl] for k, v in batch.items()}
    m = [{k: torch.from_numpy(np.ascontiguousarray(v[:, sl] if k == 'sample' else v[sl])).cuda() for k, v in ms.items()}
         for ms in masks]
    return b, m


def _run(batch, masks, reducer):
    from tacotron_amd.config import Config
    from tacotron_amd.model import Tacotron
    c = Config()
    c.r, c.vocab_size = R, V
    m = Tacotron(c, batch, train=True, seed=1, reducer=reducer)
    losses = []
    for ms in masks:
        m.step(lr=1e-3, masks=ms)
        losses.append(float(m.loss))
    torch.cuda.synchronize()
    m.check()
    return m.params.flat.cpu().numpy(), losses, float(m.global_gradient_norm)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from tacotron_amd.dist import GradReducer, init_from_env
    torch.cuda.set_device(0)
    r, w, _ = init_from_env('gloo')
    assert (r, w) == (rank, world)
    batch, masks = _case()
    half = B // world
    b, m = _slice(batch, masks, slice(rank * half, (rank + 1) * half))
    red = GradReducer(bucket_floats=1 << 20)   # several buckets per segment
    assert red.world == 2 and red.rank == rank
    params, losses, gn = _run(b, m, red)
    q.put((rank, params, losses, gn))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_process_step_matches_full_batch(built_lib):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=480) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    batch, masks = _case()
    b, m = _slice(batch, masks, slice(0, B))
    full, full_losses, full_gn = _run(b, m, None)
    p0, p1 = res[0][1], res[1][1]
    assert np.array_equal(p0, p1), 'replicas diverged'
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]          # same reduced loss / global norm on both ranks
    err = np.linalg.norm(p0 - full) / np.linalg.norm(full)
    # the first update is sign-like (Adam, zero slots): compare the accumulated parameter MOTION, not just the parameters
    from tacotron_amd.params import ParamBuffer
    init = ParamBuffer(built_lib.make_shape(B, TT, TD, R, V), 'cpu').init_(1).flat.numpy()
    move = np.linalg.norm((p0 - init) - (full - init)) / np.linalg.norm(full - init)
    print('  2 ranks vs full batch after %d steps: params rel-L2 %.2e, update rel-L2 %.2e; losses %s vs %s; gnorm %.4f vs %.4f'
          % (STEPS, err, move, res[0][2], full_losses, res[0][3], full_gn))
    assert err < 1e-6
    assert move < 2e-2   # Adam's m/sqrt(v) amplifies gradient rounding on near-zero gradient entries
    for a, f in zip(res[0][2], full_losses):
        assert abs(a - f) <= 1e-5 * abs(f)
    assert abs(res[0][3] - full_gn) <= 1e-4 * full_gn


def test_comm_standin_coresidency_with_decoder_bptt(built_lib):
    """An RCCL-footprint stand-in (64 workgroups x 256 threads x 64 KB LDS, spinning) on the high-priority communication stream
    while taco_backward runs at S1, enqueued where GradReducer enqueues the post-net segment's all-reduce.  The segment is
    announced AFTER the BPTT kernel, so a collective can never compete with that persistent launch: the BPTT runs at its solo
    time, no exchange time-out, and the stand-in's enqueue point lies behind the BPTT kernel."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'dp_coresidency', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'dp_coresidency.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.measure(spin_us=2000, reps=3, verbose=True)
    solo = res['solo']['bptt_ms']
    for name, r in res.items():
        assert r['err'] == [0, 0], (name, r)
    d = res['default: post-net segment announced after the BPTT kernel']
    assert d['bptt_ms'] <= 1.1 * solo
    assert d['spin_start_after_bwd_start_ms'] >= d['bptt_ms']          # its enqueue point lies behind the BPTT kernel


@pytest.mark.timeout(900)
def test_bench_self_launches_n_ranks_from_the_plain_command(built_lib):
    """`python bench.py --gpus 2` must start by itself: no WORLD_SIZE in the environment -> bench.py re-executes
    itself under torch.distributed.run, one process per rank.  On a 1-GPU box `--rehearse-shared-device` puts both ranks on GPU 0
    over gloo at a toy shape, so the self-launch and every N > 1 branch of the bench (process group, GradReducer on a communication
    stream, barriers, max-over-ranks timing, per-rank decoder times, ONE JSON line from rank 0) execute here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rehearse-shared-device', '--steps', '3',
                        '--warmup', '1'], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    print('  rehearsal line: n_gpus %d, %.3f ms/step, allreduce backend %s world %d, per_rank %s'
          % (j['n_gpus'], j['ms_per_step'], j['allreduce']['backend'], j['allreduce']['world'], j['per_rank']))
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['scaling'] == 'weak' and 'rehearsal' in j
    assert j['config']['global_batch'] == 2 * 4 and j['config']['parallelism'] == 'dp2'
    assert j['allreduce']['world'] == 2 and j['allreduce']['backend'] == 'gloo' and len(j['allreduce']['segments']) == 5
    assert [p['rank'] for p in j['per_rank']] == [0, 1] and all(p['us_per_decoder_step_fwd'] > 0 for p in j['per_rank'])
    assert j['value'] > 0 and np.isfinite(j['final_loss'])
