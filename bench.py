"""bench.py -- mel-frames/sec of the Tacotron train step (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = forward + backward + (gradient all-reduce) + global-norm clip + Adam on one synthetic Nancy-shaped batch
per GPU (configs[1] of BASELINE.json: B=32, r=2, Tt=200, Td=180 => 360 mel frames per utterance, scheduled-sampling
0.5, dropout 0.5), including the per-step generation of the dropout / sampling masks.  Inputs are resident in HBM
before the timed region.  value = N * 32 * 360 / (max-over-ranks time per step).  Weak scaling (32 utterances per GPU).

Extra objects on the JSON line:
  roofline     -- for the dominant kernel (persistent decoder fwd/bwd): algorithmic fp32 FLOPs per launch / average
                  launch duration measured with HIP events on the launch stream inside the timed region, against the
                  157.3 TFLOP/s fp32 (vector == f32-MFMA) peak.  See DESIGN.md for why that kernel is latency bound.
  cpu_baseline -- the CPU restatement (oracle/taco_torch.py, fp32, torch-CPU GEMMs; NOT TensorFlow -- TF 1.2 cannot
                  be installed) timed on this box's host cores on the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def decoder_flops(B, Tt, Td, r):
    """forward FLOPs of the decoder per step per row (multiply-add = 2), times B*Td."""
    R80 = 80 * r
    per_step = (2 * (80 * 256 + 256 * 128) + 2 * 384 * 256 + 6 * (512 * 512 + 512 * 256) + 2 * 256 * R80 +
                2 * R80 * 256 + 2 * (R80 + 256) * 256 + Tt * 1536)
    return float(B) * Td * per_step


def cpu_baseline(B, Tt, Td, r, V, steps=2):
    """fp32 CPU restatement, forward + backward + clip + Adam, same shapes/seeds.  Test infrastructure used as a
    reported baseline only."""
    import numpy as np

    # many tiny GEMMs inside 180-step Python loops: torch-CPU is fastest with a moderate thread count (8 threads here:
    # 6 s/step; 128 threads on the GPU box: 21 s/step), so cap it and report the count actually used
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    from oracle import taco_numpy as on
    from oracle import taco_torch as ot
    from tacotron_amd.data import synthetic_batch
    b = synthetic_batch(B, Tt, Td, r, V)
    p = {k: torch.tensor(v, requires_grad=True) for k, v in on.init_params(V, r, seed=0, dtype=np.float32).items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    inp = {'text': b['text'].long(), 'text_length': b['text_length'].long(), 'mel': b['mel'], 'stft': b['stft']}
    g = torch.Generator().manual_seed(0)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        masks = {'enc_keep1': torch.randint(0, 2, (B, Tt, 256), generator=g).float(),
                 'enc_keep2': torch.randint(0, 2, (B, Tt, 128), generator=g).float(),
                 'dec_keep1': torch.randint(0, 2, (B, Td, 256), generator=g).float(),
                 'dec_keep2': torch.randint(0, 2, (B, Td, 128), generator=g).float(),
                 'sample': torch.randint(0, 2, (Td, B), generator=g).float()}
        s2s, out, _, _ = ot.forward(p, inp, r, Td, True, masks)
        loss = ot.loss_fn(s2s, out, inp['mel'], inp['stft'])
        loss.backward()
        grads = {k: t.grad for k, t in p.items()}
        ot.clip_adam_step(p, grads, m, v, it + 1, 5e-4)
        for t in p.values():
            t.grad = None
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    sec = sorted(times)[len(times) // 2]
    return {'value': B * Td * r / sec, 'unit': 'mel-frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'full workload (B=%d,Tt=%d,Td=%d,r=%d), median of %d steps after 1 warm-up, %.2f s/step; '
                      'CPU restatement oracle/taco_torch.py (fp32), not TensorFlow' % (B, Tt, Td, r, steps, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--text-len', type=int, default=200)
    ap.add_argument('--dec-steps', type=int, default=180)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-inference', action='store_true', help='skip the inference timing (clean per-kernel profiles of the train step)')
    ap.add_argument('--speakers', type=int, default=1, help='>1: VCTK-shaped multi-speaker model (BASELINE configs[4])')
    args = ap.parse_args()

    from tacotron_amd import lib
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.dist import GradReducer, init_from_env
    from tacotron_amd.model import Tacotron

    rank, world, local = init_from_env()
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the product path)'
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    torch.cuda.set_device(local)

    c = Config()
    c.r, c.vocab_size, c.num_speakers = 2, 60, args.speakers
    B, Tt, Td = args.batch, args.text_len, args.dec_steps
    batch = synthetic_batch(B, Tt, Td, c.r, c.vocab_size, seed=1234, rank=rank, num_speakers=args.speakers)
    reducer = GradReducer() if world > 1 else None
    model = Tacotron(c, batch, train=True, seed=0, reducer=reducer)   # same init on every rank (seed 0); the mask
    #                                                                   streams are offset by the reducer's rank (model.py)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        model.step()
    torch.cuda.synchronize()
    lib.profile_read(0), lib.profile_read(1)
    lib.profile_enable(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    lib.profile_enable(False)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device='cuda')
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    sec_per_step = float(elapsed.item()) / args.steps
    fwd_ms = lib.profile_read(0)
    bwd_ms = lib.profile_read(1)
    loss = float(model.loss)

    # ---- inference (BASELINE configs[3]: prompt -> mel -> linear, Tt=140 as data_input.MAX_TEXT_LEN, always Td steps) ----
    infer = None
    if rank == 0 and world == 1 and not args.no_inference:
        ci = Config()
        ci.r, ci.vocab_size, ci.num_speakers = 2, 60, args.speakers
        ci.max_decode_iter = Td
        infer = {}
        for Bi in (1, 32):
            bi = synthetic_batch(Bi, 140, Td, ci.r, ci.vocab_size, seed=77, min_len=40, num_speakers=args.speakers)
            mi = Tacotron(ci, bi, train=False, params=None, seed=0)
            for _ in range(2):
                mi.run()
            torch.cuda.synchronize()
            ti = time.perf_counter()
            n_it = 5
            for _ in range(n_it):
                mi.run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - ti) / n_it * 1e3
            infer['B%d' % Bi] = {'ms_per_batch': ms, 'mel_frames_per_s': Bi * Td * ci.r / (ms * 1e-3)}
            del mi
        infer['shape'] = 'Tt=140, Td=%d steps (r=2 -> %d frames/utt), full forward incl. post-net + linear' % (Td, Td * 2)

    if rank == 0:
        frames = world * B * Td * c.r
        fa = sum(fwd_ms) / max(1, len(fwd_ms))
        ba = sum(bwd_ms) / max(1, len(bwd_ms))
        dom, dom_ms = ('decoder_bwd_kernel', ba) if ba >= fa else ('decoder_fwd_kernel', fa)
        # algorithmic FLOPs of ONE launch: forward = SURVEY §8(d) decoder figure; the backward kernel does the
        # transposed mat-vecs + attention backward = the same count again (weight gradients are separate GEMMs).
        flops = decoder_flops(B, Tt, Td, c.r)
        achieved = flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        res = {
            'metric': 'mel-frames/sec (train step, batch=32 r=2)', 'value': frames / sec_per_step, 'unit': 'mel-frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': sec_per_step * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'Nancy-shaped train step (BASELINE configs[1]): B=%d/GPU, r=%d, Tt=%d chars, Td=%d steps '
                                   '(%d mel frames/utt), sched-sampling 0.5, dropout 0.5, V=60, speakers=%d, fwd+bwd+clip+Adam'
                                   % (B, c.r, Tt, Td, Td * c.r, args.speakers),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': 157.3, 'unit': 'TFLOP/s',
                         'frac': achieved / 157.3, 'traffic': traffic, 'kernel': dom, 'avg_ms': dom_ms,
                         'launches_timed': len(bwd_ms if dom.startswith('decoder_bwd') else fwd_ms),
                         'flops_per_launch': flops,
                         'note': 'persistent per-row recurrence, fp32 FMA; latency/L2-stream bound (DESIGN.md)'},
            'kernels_ms': {'decoder_fwd_kernel': fa, 'decoder_bwd_kernel': ba,
                           'us_per_decoder_step_fwd': fa * 1e3 / Td, 'us_per_decoder_step_bwd': ba * 1e3 / Td},
            'final_loss': loss,
        }
        if infer:
            res['inference'] = infer
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(B, Tt, Td, c.r, c.vocab_size)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
