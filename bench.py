"""bench.py -- mel-frames/sec of the Tacotron train step (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: re-executes itself under
                                                           `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus 2 --rehearse-shared-device      (N ranks on GPU 0 over gloo at a toy shape: exercises the self-launch and
                                                           the whole N > 1 code path on a 1-GPU box; not a measurement)

A "step" = forward + backward + (gradient all-reduce) + global-norm clip + Adam on one synthetic Nancy-shaped batch
per GPU (configs[1] of BASELINE.json: B=32, r=2, Tt=200, Td=180 => 360 mel frames per utterance, scheduled-sampling
0.5, dropout 0.5), including the per-step generation of the dropout / sampling masks.  Inputs are resident in HBM
before the timed region.  value = N * 32 * 360 / (max-over-ranks time per step).  Weak scaling (32 utterances per GPU).

Extra objects on the JSON line:
  roofline     -- for the dominant kernel (persistent decoder fwd/bwd): algorithmic fp32 FLOPs per launch / average
                  launch duration measured with HIP events on the launch stream inside the timed region, against the
                  157.3 TFLOP/s fp32 (vector == f32-MFMA) peak.  See DESIGN.md for why that kernel is latency bound.
  rooflines    -- the same per family: dominant kernel, MFMA GEMM family (events around every launch, separate untimed
                  pass), whole step (SURVEY 8d FLOPs / ms_per_step), bi-GRU recurrences, inference B=1.
  s2 / vctk    -- the S2 envelope (Td=500 = 1000 mel frames) and the 109-speaker configuration, same step, 1 GPU.
  box / ab     -- `box`: what this box's fabric measures (granule hop latency same-XCD / cross-XCD, L2-hit and all-miss load latency,
                  one CU's stream bandwidth, latency-bound shader clock); `ab`: same-box, same-run A/B of the big-GEMM arithmetic
                  (bf16x3 vs fp32 MFMA) and of the decoder build (product vs previous form, libtaco_prevdec.so).
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

# HIP reads this at its FIRST API call (torch.cuda.is_available() is one): before torch is imported, not after (tacotron_amd/lib.py)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402


def decoder_flops(B, Tt, Td, r):
    """forward FLOPs of the decoder per step per row (multiply-add = 2), times B*Td."""
    R80 = 80 * r
    per_step = (2 * (80 * 256 + 256 * 128) + 2 * 384 * 256 + 6 * (512 * 512 + 512 * 256) + 2 * 256 * R80 +
                2 * R80 * 256 + 2 * (R80 + 256) * 256 + Tt * 1536)
    return float(B) * Td * per_step


def cpu_baseline(B, Tt, Td, r, V, steps=5):
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
    return {'value': B * Td * r / sec, 'unit': 'mel-frames/s', 'cores': torch.get_num_threads(), 'nproc': os.cpu_count(),
            'cores_note': 'threads actually used; more are slower on this graph of tiny GEMMs inside 180-step loops (128 threads: 21 s/step)',
            'kind': 'port',
            'sample': 'full workload (B=%d,Tt=%d,Td=%d,r=%d), median of %d steps after 1 warm-up, %.2f s/step; '
                      'CPU restatement oracle/taco_torch.py (fp32), not TensorFlow' % (B, Tt, Td, r, steps, sec)}


def model_flops(B, Tt, Td, r):
    """algorithmic forward FLOPs (multiply-add = 2); a train step is 3x."""
    enc = 7110656.0                       # per text position: pre_net, conv bank, projections, highways, bi-GRU, memory layer
    dec = 3039232.0 + 1536.0 * Tt         # per decoder step and row (r = 2)
    post = 3633664.0                      # per mel frame
    return B * (Tt * enc + Td * dec + Td * r * post)


def source_hash():
    """sha1 over the kernel sources: ties a PMC summary under profiles/ to the build it was measured on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'tacotron_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:12]


def time_steps(model, steps, warmup, barrier, world):
    from tacotron_amd import lib
    for _ in range(warmup):
        model.step()
    torch.cuda.synchronize()
    lib.profile_read(0), lib.profile_read(1)
    lib.profile_enable(3)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    lib.profile_enable(0)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device='cuda')
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    fwd_ms, bwd_ms = lib.profile_read(0), lib.profile_read(1)
    return float(elapsed.item()) / steps, fwd_ms, bwd_ms


def family_profile(model, steps=3):
    """Separate, untimed pass: every MFMA GEMM-family launch and every bi-GRU recurrence bracketed by HIP events on its
    launch stream (taco_profile_enable bits 2, 3).  Returns per-step sums."""
    from tacotron_amd import lib
    lib.profile_read(2), lib.profile_read(3)
    lib.profile_enable(0b11100)   # bit 4: no side stream during this pass -- every launch is timed by itself
    for _ in range(steps):
        model.step()
    torch.cuda.synchronize()
    lib.profile_enable(0)
    gm, gf = lib.profile_read(2, with_flops=True)
    rm, _ = lib.profile_read(3, with_flops=True)
    return {'gemm_ms': sum(gm) / steps, 'gemm_flops': sum(gf) / steps, 'gemm_launches': len(gm) / steps,
            'bigru_ms': sum(rm) / steps, 'bigru_launches': len(rm) / steps}


def fp32_gemm_4096():
    """Two reference points for the GEMM-family fraction, measured in this run on a 4096^3 fp32 GEMM (two full waves of 128 x 128
    tiles, 128 k-tiles each), 5 launches after 2 warm-up launches each: the library's own NN kernel and the vendor BLAS (torch.mm,
    hipBLASLt).  A ~10 ms burst like this runs ~10-15 % below what a multi-second loop reaches on the same box (139 and 150 TFLOP/s,
    profiles/r04_gemm_steady.txt): the clocks ramp over tens of milliseconds, and the GEMM phases of a train step are bursts too."""
    from tacotron_amd import lib
    n = 4096
    A = torch.randn(n, n, device='cuda')
    W = torch.randn(1, n, n, device='cuda') * 0.05
    C = torch.empty(n, n, device='cuda')

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 2.0 * n ** 3 / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e12
    ours = timed(lambda: lib.conv_gemm(A, W, C, n, n, n, taps=1, T=n, pad_l=0, act=0))
    try:
        blas = timed(lambda: torch.mm(A, W[0], out=C))
    except Exception:   # noqa: BLE001 -- context only
        blas = None
    return ours, blas


def ab_gemm(model, rounds=3, steps=8):
    """In-run A/B of the big-GEMM arithmetic: the bf16x3 form (three-way exact operand split on the bf16 matrix
    pipe, the default) against the fp32 MFMA form of rounds 2-4, alternated `rounds` times in THIS process on THIS box -- the switch
    is read at every launch (TACO_GEMM2_BF16X)."""
    out = {'bf16x3': [], 'fp32_mfma': []}
    prev = os.environ.get('TACO_GEMM2_BF16X')
    try:
        for _ in range(rounds):
            for name, v in (('bf16x3', '1'), ('fp32_mfma', '0')):
                os.environ['TACO_GEMM2_BF16X'] = v
                for _ in range(2):
                    model.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    model.step()
                torch.cuda.synchronize()
                out[name].append((time.perf_counter() - t0) / steps * 1e3)
    finally:
        if prev is None:
            os.environ.pop('TACO_GEMM2_BF16X', None)
        else:
            os.environ['TACO_GEMM2_BF16X'] = prev
    med = lambda x: sorted(x)[len(x) // 2]   # noqa: E731
    return {'what': 'S1 train step, ms: big GEMMs on the bf16x3 form (default) vs the fp32 MFMA form of rounds 2-4, alternated in this run',
            'ms_per_step': {k: [round(x, 3) for x in v] for k, v in out.items()},
            'median_ms': {k: med(v) for k, v in out.items()}, 'gain_ms': med(out['fp32_mfma']) - med(out['bf16x3'])}


def ab_env(model, var, on, off, names, what, rounds=3, steps=8):
    """In-run A/B of a launch-time switch of the library (read at every launch / call): `var` = `on` vs `off`, alternated in THIS process."""
    out = {names[0]: [], names[1]: []}
    prev = os.environ.get(var)
    try:
        for _ in range(rounds):
            for name, v in ((names[0], on), (names[1], off)):
                os.environ[var] = v
                for _ in range(2):
                    model.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    model.step()
                torch.cuda.synchronize()
                out[name].append((time.perf_counter() - t0) / steps * 1e3)
    finally:
        if prev is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = prev
    med = lambda x: sorted(x)[len(x) // 2]   # noqa: E731
    return {'what': what, 'ms_per_step': {k: [round(x, 3) for x in v] for k, v in out.items()},
            'median_ms': {k: med(v) for k, v in out.items()}, 'gain_ms': med(out[names[1]]) - med(out[names[0]])}


def ab_decoder(rounds=3):
    """In-run A/B of decoder builds: the product library against libtaco_prevdec.so (the same sources with the previous decoder
    form: -DTACO_NO_RS -DTACO_NO_POLL128 -DTACO_NO_SHADOW -DTACO_NO_GROUPED_FANDQ -DTACO_NO_UNIPOLL -DTACO_NO_TANH_SPLIT), alternated `rounds` times on this box
    (one short subprocess each: tools/dec_quick.py --json, S1 shape)."""
    import subprocess
    libs = {'current': os.path.join(ROOT, 'tacotron_amd', 'libtaco_hip.so'), 'previous_form': os.path.join(ROOT, 'tacotron_amd', 'libtaco_prevdec.so')}
    if not all(os.path.exists(p) for p in libs.values()):
        return None
    runs = {k: [] for k in libs}
    for _ in range(rounds):
        for k, path in libs.items():
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dec_quick.py'), '--time-only', '--json'],
                                   env=dict(os.environ, TACO_LIB=path), capture_output=True, text=True, timeout=120)
                line = [x for x in r.stdout.splitlines() if x.startswith('{')]
                if line:
                    runs[k].append(json.loads(line[-1]))
            except Exception:   # noqa: BLE001 -- a diagnostic, never fatal for the bench line
                pass
    if not all(runs.values()):
        return None
    med = lambda k, f: sorted(x[f] for x in runs[k])[len(runs[k]) // 2]   # noqa: E731
    return {'what': 'S1 train step: product build vs the round-3 decoder form (column sums by wave / 8-byte polls / no poll-shadow work / '
                    'per-lane poll loops / tanh in its sum form), %d alternations on this box' % rounds,
            **{k: {'ms_per_step': med(k, 'ms_per_step'), 'us_per_decoder_step_fwd': med(k, 'us_per_decoder_step_fwd'),
                   'us_per_decoder_step_bwd': med(k, 'us_per_decoder_step_bwd'), 'runs': len(runs[k])} for k in libs}}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: re-execute this script under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (the form the
    driver uses), one rank per GPU over RCCL; rank 0's ONE JSON line passes through on stdout.  Returns the exit code."""
    import socket
    import subprocess
    if not args.rehearse_shared_device:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write('bench.py: --gpus %d but this node shows %d GPU(s) (use --rehearse-shared-device to run %d ranks on GPU 0 '
                             'over gloo at a toy shape)\n' % (args.gpus, have, args.gpus))
            return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on these hosts (RCCL / device-tensor sharing)
    env.setdefault('GPU_MAX_HW_QUEUES', '8')            # before any rank touches HIP (tacotron_amd/lib.py)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write('bench.py: self-launch: %s\n' % ' '.join(cmd))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='utterances per GPU (default 32; 4 with --rehearse-shared-device)')
    ap.add_argument('--text-len', type=int, default=None, help='padded text length Tt (default 200; 24 in the rehearsal)')
    ap.add_argument('--dec-steps', type=int, default=None, help='decoder steps Td (default 180; 10 in the rehearsal)')
    ap.add_argument('--rehearse-shared-device', action='store_true',
                    help='run the N ranks on GPU 0 over gloo at a toy shape (RCCL refuses two ranks on one device): a rehearsal of the '
                         'self-launch and of every N > 1 branch on a 1-GPU box, NOT a measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-inference', action='store_true', help='skip the inference timing (clean per-kernel profiles of the train step)')
    ap.add_argument('--no-extras', action='store_true', help='skip the S2 (Td=500) and VCTK (109 speakers) legs and the family profile')
    ap.add_argument('--speakers', type=int, default=1, help='>1: VCTK-shaped multi-speaker model (BASELINE configs[4])')
    args = ap.parse_args()
    toy = args.rehearse_shared_device
    args.batch = args.batch or (4 if toy else 32)
    args.text_len = args.text_len or (24 if toy else 200)
    args.dec_steps = args.dec_steps or (10 if toy else 180)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    # The contract is ONE JSON line on stdout.  Libraries underneath write to file descriptor 1 on their own (RCCL prints a five-line
    # version banner at communicator creation): everything up to the result line goes to stderr, then stdout is put back.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    from tacotron_amd import lib
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.dist import GradReducer, init_from_env
    from tacotron_amd.model import Tacotron

    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the product path)'
    if toy:
        torch.cuda.set_device(0)
        rank, world, local = init_from_env('gloo')   # every rank on GPU 0; DEVICE tensors travel over gloo
        local = 0
    else:
        rank, world, local = init_from_env()
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    torch.cuda.set_device(local)

    c = Config()
    c.r, c.vocab_size, c.num_speakers = 2, 60, args.speakers
    B, Tt, Td = args.batch, args.text_len, args.dec_steps
    # TACO_FORCE_DIST=1: take the whole distributed path (RCCL process group, communication stream, segment events, bucketed
    # collectives, barriers) at world size 1 too -- `torchrun --nproc-per-node 1 bench.py --gpus 1` then executes every branch
    # the 8-GPU run takes
    forced = torch.distributed.is_initialized() and world == 1
    reducer = GradReducer(force=forced) if (world > 1 or forced) else None

    def barrier():
        if world > 1 or forced:
            torch.distributed.barrier()

    def make_model(Td_, speakers, B_=None):
        cc = Config()
        cc.r, cc.vocab_size, cc.num_speakers = 2, 60, speakers
        batch = synthetic_batch(B_ or B, Tt, Td_, cc.r, cc.vocab_size, seed=1234, rank=rank, num_speakers=speakers)
        # same initial parameters on every rank (seed 0); the mask streams are offset by the reducer's rank (model.py)
        return Tacotron(cc, batch, train=True, seed=0, reducer=reducer)

    model = make_model(Td, args.speakers)
    sec_per_step, fwd_ms, bwd_ms = time_steps(model, args.steps, args.warmup, barrier, world)
    loss = float(model.loss)
    model.check()
    train_cluster, train_mode = lib.last_cluster(0), lib.decoder_mode()
    allreduce = None
    if reducer is not None and rank == 0:
        # separate untimed pass: events around every segment's collectives on the communication stream
        reducer.timing = True
        for _ in range(3):
            model.step()
        seg_us = reducer.segment_times_us()
        reducer.timing = False
        allreduce = {'backend': torch.distributed.get_backend(), 'world': world, 'bucket_floats': reducer.bucket,
                     'comm_stream_priority': reducer.comm_priority,
                     'segments': [dict(d, us=seg_us[d['segment']]) for d in reducer.describe(model)],
                     'note': 'completion order inside taco_backward; us = segment final (its event reached on the communication '
                             'stream) -> its last collective done, median of 3 untimed steps'}
    elif reducer is not None:
        for _ in range(3):
            model.step()   # (collectives need every rank)
    fam = ab = None
    if rank == 0 and world == 1 and not args.no_extras:
        fam = family_profile(model)
        ab = {'gemm': ab_gemm(model),
              'weight_images': ab_env(model, 'TACO_GEMM2_BSPLIT', '1', '0', ('image_form', 'split_in_registers'),
                                      'S1 train step, ms: weight operand of the big NN GEMMs read from pre-split bf16 plane images (default, round 6) '
                                      'vs split in registers in every wave (round 5), alternated in this run; results are bit-identical'),
              'tail_events': ab_env(model, 'TACO_TAIL_EVENTS', '1', '0', ('stop_event_on_the_launch', 'recorded_marker'),
                                    'S1 train step, ms: cross-stream forks / joins wait for the stop event riding on the producing '
                                    "stream's last launch (default, round 6 late) vs a recorded marker packet between two kernels of the "
                                    'producing stream (rounds 1-6), alternated in this run; same kernels, same order')}
    del model
    torch.cuda.empty_cache()

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
            # median of per-call times: one call in ~100 stalls for 80-90 ms on the HOST (profiles/r03_inference_outlier_probe.txt),
            # which a mean over a handful of calls turns into a 2-3x "slowdown"
            calls = []
            for _ in range(9):
                ti = time.perf_counter()
                mi.run()
                torch.cuda.synchronize()
                calls.append((time.perf_counter() - ti) * 1e3)
            ms = sorted(calls)[len(calls) // 2]
            mi.check()
            infer['B%d' % Bi] = {'ms_per_batch': ms, 'mel_frames_per_s': Bi * Td * ci.r / (ms * 1e-3),
                                 'ms_per_call': [round(x, 3) for x in calls],
                                 'decoder_cluster_width': lib.last_cluster(0), 'placement_census': mi.placement_census(),
                                 'gflop': model_flops(Bi, 140, Td, ci.r) / 1e9,
                                 'tflops': model_flops(Bi, 140, Td, ci.r) / (ms * 1e-3) / 1e12}
            del mi
        infer['shape'] = 'Tt=140, Td=%d steps (r=2 -> %d frames/utt), full forward incl. post-net + linear' % (Td, Td * 2)

    # ---- the other measured configurations (rank 0 of a 1-GPU run only; the driver's scaling runs skip them) ----
    s2 = vctk = b64 = None
    if rank == 0 and world == 1 and not args.no_extras:
        def leg(Td_, speakers, steps=8, warmup=3, B_=None):
            m = make_model(Td_, speakers, B_)
            # best of two timed passes: about one host call in a hundred stalls for 80-90 ms (profiles/r03_inference_outlier_probe.txt),
            # and one such stall inside an 8-step pass reads as +1 ms per step (seen once in round 6: 8.53 instead of 7.45)
            sec, f, b_ = time_steps(m, steps, warmup, barrier, 1)
            sec2, f2, b2_ = time_steps(m, steps, 0, barrier, 1)
            if sec2 < sec:
                sec, f, b_ = sec2, f2, b2_
            m.check()
            fa_, ba_ = sum(f) / max(1, len(f)), sum(b_) / max(1, len(b_))
            del m
            torch.cuda.empty_cache()
            return {'ms_per_step': sec * 1e3, 'mel_frames_per_s': (B_ or B) * Td_ * 2 / sec, 'steps': steps, 'warmup': warmup, 'passes': 'best of 2',
                    'decoder_fwd_ms': fa_, 'decoder_bwd_ms': ba_, 'us_per_decoder_step_fwd': fa_ * 1e3 / Td_,
                    'us_per_decoder_step_bwd': ba_ * 1e3 / Td_, 'train_gflop_per_step': 3 * model_flops(B_ or B, Tt, Td_, 2) / 1e9}
        if Td != 500:
            s2 = leg(500, args.speakers)
            s2['workload'] = 'S2 envelope: B=%d, Tt=%d, Td=500 (1000 mel frames/utt), r=2' % (B, Tt)
        if args.speakers == 1 and B == 32:
            # off-metric (BASELINE quotes batch = 32 per GPU): a batch of 64 per GPU on the same kernels -- the decoder runs it as two
            # consecutive launches of 32 rows (rounds 1-5: the decoder.hip fallback, 2.35 x slower per decoder step)
            b64 = leg(Td, 1, B_=64)
            b64['workload'] = 'OFF-METRIC: B=64 per GPU, Tt=%d, Td=%d, r=2 (decoder3.hip in two launches of 32 rows)' % (Tt, Td)
        if args.speakers == 1:
            vctk = leg(Td, 109)
            vctk['workload'] = 'VCTK-shaped (BASELINE configs[4], 1 GPU): 109 speakers, B=%d, Tt=%d, Td=%d, r=2' % (B, Tt, Td)

    fa = sum(fwd_ms) / max(1, len(fwd_ms))
    ba = sum(bwd_ms) / max(1, len(bwd_ms))
    per_rank = None
    if world > 1:
        # every rank's own decoder launch times (HIP events on its launch stream): rank r fills slot r, SUM gathers them
        slots = torch.zeros(world, 2, dtype=torch.float64, device='cuda')
        slots[rank, 0], slots[rank, 1] = fa * 1e3 / Td, ba * 1e3 / Td
        torch.distributed.all_reduce(slots, op=torch.distributed.ReduceOp.SUM)
        per_rank = [{'rank': i, 'us_per_decoder_step_fwd': float(slots[i, 0]), 'us_per_decoder_step_bwd': float(slots[i, 1])}
                    for i in range(world)]
    if rank == 0:
        frames = world * B * Td * c.r
        dom, dom_ms = ('decoder3_bwd_kernel', ba) if ba >= fa else ('decoder3_fwd_kernel', fa)
        # algorithmic FLOPs of ONE launch: forward = SURVEY 8(d) decoder figure; the backward kernel does the
        # transposed mat-vecs + attention backward = the same count again (weight gradients are separate GEMMs).
        flops = decoder_flops(B, Tt, Td, c.r)
        achieved = flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # HBM-side traffic per launch from the PMC passes of tools/profile_round.sh -- only if they were collected on THIS
        # build of the kernels (source hash), else null
        traffic = traffic_step = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get('build') == source_hash():
                    traffic = pj.get(dom, {}).get('hbm_bytes_per_launch')
                    st = pj.get('step')
                    if st:
                        traffic_step = {'bytes': st['step_bytes'], 'algorithmic_bytes': st['algorithmic_step_bytes'],
                                        'ratio': st['ratio'],
                                        'families': {k: v['bytes'] for k, v in st['families'].items()}}
            except Exception:
                traffic = traffic_step = None
        PEAK = 157.3
        OWN_ROOF = 2516.0 / 6.0
        step_flops = 3 * model_flops(B, Tt, Td, c.r)
        rooflines = [
            # (`bound` names the roof the kernel is priced against -- the contract's enum is hbm | mfma; `limiter` says what
            #  actually binds: neither roof, the kernel is a latency-bound recurrence)
            {'what': dom, 'bound': 'mfma', 'bound_actual': 'latency', 'limiter': 'latency', 'achieved': achieved, 'peak': PEAK,
             'unit': 'TFLOP/s', 'frac': achieved / PEAK,
             'avg_ms': dom_ms, 'us_per_decoder_step': dom_ms * 1e3 / Td, 'flops_per_launch': flops, 'traffic': traffic,
             'traffic_step': traffic_step,
             'traffic_source': 'profiles/pmc_latest.json (rocprofv3 --pmc passes of tools/profile_round.sh; a committed constant, '
                               'used only when its source hash equals this build -- not measured in this run)' if traffic else None,
             'note': 'persistent recurrence (decoder3.hip: 8 clusters x 32 workgroups x 4 rows, register-resident weights): %d strictly '
                     'sequential steps x 8-9 dependent exchange rounds, no MFMA, not HBM bound; the fp32 peak is quoted for scale '
                     'only (DESIGN.md 5)' % Td},
            {'what': 'whole train step', 'bound': 'mfma', 'achieved': step_flops / sec_per_step / 1e12, 'peak': PEAK,
             'unit': 'TFLOP/s', 'frac': step_flops / sec_per_step / 1e12 / PEAK, 'flops_per_step': step_flops,
             'ms_per_step': sec_per_step * 1e3},
        ]
        if fam and fam['gemm_ms'] > 0:
            g = fam['gemm_flops'] / (fam['gemm_ms'] * 1e-3) / 1e12
            ours4k, blas4k = fp32_gemm_4096()
            rooflines.insert(1, {'what': 'MFMA GEMM family (conv_gemm + gemm_tn + fused highway launches)', 'bound': 'mfma',
                                 'achieved': g, 'peak': PEAK, 'unit': 'TFLOP/s', 'frac': g / PEAK,
                                 # the big launches of the family form each fp32 product from six v_mfma_f32_32x32x16_bf16: the
                                 # roof they actually run under is the dense bf16 peak / 6, not the fp32-instruction peak above
                                 'peak_own_roof_tflops': OWN_ROOF, 'frac_own_roof': g / OWN_ROOF,
                                 'own_roof_note': 'bf16 dense peak 2,516 TFLOP/s / 6 plane products per fp32 product (bf16x3 form, the default '
                                                  'of conv_gemm2 / gemm_tn); `frac` keeps the fp32-MFMA peak the contract names',
                                 'nn_kernel_4096_cubed_tflops': ours4k, 'vendor_blas_4096_cubed_tflops': blas4k,
                                 'reference_note': 'a 4096^3 fp32 GEMM on the library\'s NN kernel and on the vendor BLAS (torch.mm), 5 launches '
                                                   'each in this run: what a long, tail-free launch reaches under burst clocks -- context for '
                                                   'the family fraction, not a ceiling',
                                 'ms_per_step_summed': fam['gemm_ms'], 'flops_per_step': fam['gemm_flops'],
                                 'launches_per_step': fam['gemm_launches'],
                                 'note': 'HIP events around every launch; measured in a separate pass with the side stream switched off '
                                         '(taco_profile_enable bit 4) so that no two launches share the chip -- in the timed step ~40 % '
                                         'of these launches run on the side stream beside the main chain'})
            rooflines.append({'what': 'bi-GRU recurrences (4 launches/step)', 'bound': 'mfma', 'limiter': 'latency',
                              'ms_per_step_summed': fam['bigru_ms'], 'launches_per_step': fam['bigru_launches']})
        if infer:
            rooflines.append({'what': 'inference B=1 (whole forward)', 'bound': 'mfma', 'limiter': 'latency', 'achieved': infer['B1']['tflops'],
                              'peak': PEAK, 'unit': 'TFLOP/s', 'frac': infer['B1']['tflops'] / PEAK,
                              'ms': infer['B1']['ms_per_batch'], 'gflop': infer['B1']['gflop']})
        res = {
            'metric': 'mel-frames/sec (train step, batch=32 r=2)', 'value': frames / sec_per_step, 'unit': 'mel-frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': sec_per_step * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': 'fp32 operands, fp32 accumulation everywhere.  The big GEMM launches form each fp32 product from six bf16 plane '
                          'products on the matrix pipe (bf16x3, exact three-way operand split; the five low-order products in an accumulator '
                          'of their own) for accumulation chains of <= 2048 products (every launch of this step); deeper chains run '
                          'v_mfma_f32_32x32x2_f32.  Against fp64 that form measures 0.3-0.4 x the error of the fp32 INSTRUCTION on mixed-sign '
                          'operands (Gaussian, post-ReLU x glorot) at every depth, ~1.0 x on heavy-tailed ones; on same-signed operands with '
                          'all mantissa bits set -- its worst case -- 1.4e-7 / 1.8e-7 / 3.2e-7 rel-L2 at chains of 256 / 1024 / 2048 against '
                          '2e-8 of the fp32 instruction: inside a 4e-7 bar, asserted by tests/test_gpu_ops.py::test_bf16x3_adversarial_operands '
                          '(profiles/r06_bf16x3_acc2.txt; one accumulator, as in round 5: 1.9e-6 at 2048).  The strict fp32-instruction step '
                          'time is ab.gemm.median_ms.fp32_mfma',
            'config': {'workload': 'Nancy-shaped train step (BASELINE configs[1]): B=%d/GPU, r=%d, Tt=%d chars, Td=%d steps '
                                   '(%d mel frames/utt), sched-sampling 0.5, dropout 0.5, V=60, speakers=%d, fwd+bwd+clip+Adam'
                                   % (B, c.r, Tt, Td, Td * c.r, args.speakers),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world},
            # dominant kernel (largest share of the step); `rooflines` carries every family
            'roofline': dict(rooflines[0], kernel=dom, launches_timed=len(bwd_ms if dom.startswith('decoder3_bwd') else fwd_ms)),
            'rooflines': rooflines,
            'kernels_ms': {'decoder_fwd_kernel': fa, 'decoder_bwd_kernel': ba,
                           'us_per_decoder_step_fwd': fa * 1e3 / Td, 'us_per_decoder_step_bwd': ba * 1e3 / Td,
                           'non_decoder_critical_path_ms': sec_per_step * 1e3 - fa - ba},
            'final_loss': loss, 'build': source_hash(),
            # which box this was: the latency-bound kernels (decoder, bi-GRU: 58 % of the step) scale with the shader clock the chip
            # sustains, and boxes of one pool differ by ~10 % (round 4: the same build ran 8.86 and 9.30 ms per step)
            'box': dict({'shader_clock_ghz_latency_bound': lib.clock_probe(), 'decoder_mode': train_mode,
                         'decoder_mode_note': '0 = decoder3.hip with the XCD-local exchange (product default), 1 = decoder3.hip with the '
                                              'agent-scope exchange, 2 = decoder.hip (include/taco_hip.h taco_decoder_mode)',
                         'decoder_cluster_width_train': train_cluster},
                        **(lib.fabric_probe() if world == 1 else {})),
        }
        if per_rank:
            res['per_rank'] = per_rank
        if toy:
            res['rehearsal'] = ('%d ranks share GPU 0 and exchange device tensors over gloo at a toy shape: a rehearsal of the launch path, '
                                'the numbers are NOT measurements of the BASELINE workload' % world)
            res['data'] = 'synthetic (toy rehearsal shape)'
        if ab is not None:
            dec_ab = ab_decoder()
            if dec_ab:
                ab['decoder'] = dec_ab
            res['ab'] = ab
        if allreduce:
            res['allreduce'] = allreduce
        if s2:
            res['s2'] = s2
        if vctk:
            res['vctk'] = vctk
        if b64:
            res['b64'] = b64
        if infer:
            res['inference'] = infer
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(B, Tt, Td, c.r, c.vocab_size)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)   # (whatever is printed at teardown stays off stdout too)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
