"""Generates tests/golden/model_r{2,5}.npz from the CPU restatement (fp64).

PARITY UNPINNED: these vectors come from oracle/taco_numpy.py (cross-checked against oracle/taco_torch.py),
NOT from TensorFlow 1.2, which cannot run in this image.  Each file records the assumptions
(oracle.taco_numpy.ASSUMPTIONS) it encodes.  Parameters are regenerated from `seed` (a 27 MB dump is not a
"small fixture"); `param_checksum` guards against RNG drift.

Run from the repo root:  python -m oracle.make_golden
"""
import os

import numpy as np

from oracle import taco_numpy as on
from oracle import taco_torch as ot

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def make_case(r, V=20, B=2, Tt=9, Td=5, seed=0, num_speakers=1):
    rng = np.random.default_rng(100 + r + 1000 * (num_speakers > 1))
    p = on.init_params(V, r, seed=seed, perturb=0.3, num_speakers=num_speakers)
    text = rng.integers(1, V, size=(B, Tt)).astype(np.int32)
    tl = np.array([Tt, max(2, Tt - 3)], dtype=np.int32)[:B]
    for b in range(B):
        text[b, tl[b]:] = 0
    mel = rng.standard_normal((B, Td, 80 * r)).astype(np.float32).astype(np.float64)    # fp32-representable targets
    stft = rng.standard_normal((B, Td, 1025 * r)).astype(np.float32).astype(np.float64)
    masks = {
        'enc_keep1': rng.integers(0, 2, (B, Tt, 256)).astype(np.uint8),
        'enc_keep2': rng.integers(0, 2, (B, Tt, 128)).astype(np.uint8),
        'dec_keep1': rng.integers(0, 2, (B, Td, 256)).astype(np.uint8),
        'dec_keep2': rng.integers(0, 2, (B, Td, 128)).astype(np.uint8),
        'sample': rng.integers(0, 2, (Td, B)).astype(np.uint8),
    }
    inp = {'text': text, 'text_length': tl, 'mel': mel, 'stft': stft}
    if num_speakers > 1:
        inp['speaker'] = rng.integers(0, num_speakers, size=B).astype(np.int32)
    fm = {k: v.astype(np.float64) for k, v in masks.items()}
    s2s, out, al, enc = on.forward(p, inp, r, Td, True, fm)
    loss = on.loss_fn(s2s, out, mel, stft)
    lt, s2, o2, a2, grads = ot.loss_and_grads(p, inp, r, Td, fm)
    assert abs(loss - lt) < 1e-8 * abs(loss) and np.abs(s2s - s2).max() < 1e-10 and np.abs(out - o2).max() < 1e-10
    is2s, iout, ial, _ = on.forward(p, inp, r, Td, False)
    # attention argmax margins
    srt = np.sort(al, -1)
    margin = srt[..., -1] - srt[..., -2]
    flat = on.flatten_params(p, V, r, np.float64, num_speakers)
    d = dict(
        r=r, V=V, B=B, Tt=Tt, Td=Td, seed=seed, perturb=0.3, param_checksum=float(np.abs(flat).sum()),
        num_speakers=num_speakers, speaker=inp.get('speaker', np.zeros(B, np.int32)),
        text=text, text_length=tl, mel=mel.astype(np.float32), stft=stft.astype(np.float32),
        seq2seq_output=s2s, output=out, alignments=al, encoded=enc, loss=loss, argmax_margin=margin,
        infer_seq2seq_output=is2s, infer_output=iout, infer_alignments=ial,
        grad_names=np.array(list(grads.keys())),
        grad_norms=np.array([np.sqrt((g ** 2).sum()) for g in grads.values()]),
        grad_attention_v=grads['decoder/attention_v'], grad_in_proj_bias=grads['decoder/in_proj/bias'],
        grad_enc_bank_bn_gamma=grads['encoder/cbhg/bank_bn/gamma'],
        grad_post_dense_bias=grads['post/dense/bias'], grad_embedding=grads['embedding'],
        assumptions=np.array(on.ASSUMPTIONS),
    )
    for k, v in masks.items():
        d['mask_' + k] = v
    return d


PEAKED_SCALES = (('decoder/attention_v', 100.0), ('decoder/memory_layer/kernel', 8.0), ('decoder/query_layer/kernel', 4.0))


def make_peaked(r=2, V=40, B=4, Tt=60, Td=40, seed=11):
    """Medium fixture with PEAKED attention: attention_v / memory_layer / query_layer are scaled so that the softmax is
    dominated by one memory position on most steps (max alpha > 0.9 on ~2/3 of the (b,t), 24 distinct arg-max positions),
    i.e. the arg-max test is meaningful.  Energies are O(10^3), so fp32 rounding of the energy sum
    moves alignments by up to ~3e-5 (NumPy fp32 vs fp64 of this very file): the alignment tolerance of THIS fixture is 1e-4,
    every arg-max margin is >= 1e-2.  Floats are stored as fp32 (targets are fp32-representable; expected outputs rounded)."""
    p = on.init_params(V, r, seed=seed, perturb=0.3)
    for k, sc in PEAKED_SCALES:
        p[k] = p[k] * sc
    rng = np.random.default_rng(777)
    text = rng.integers(1, V, size=(B, Tt)).astype(np.int32)
    tl = np.array([60, 47, 33, 21], dtype=np.int32)[:B]
    for b in range(B):
        text[b, tl[b]:] = 0
    mel = rng.standard_normal((B, Td, 80 * r)).astype(np.float32).astype(np.float64)
    stft = rng.standard_normal((B, Td, 1025 * r)).astype(np.float32).astype(np.float64)
    masks = {
        'enc_keep1': rng.integers(0, 2, (B, Tt, 256)).astype(np.uint8),
        'enc_keep2': rng.integers(0, 2, (B, Tt, 128)).astype(np.uint8),
        'dec_keep1': rng.integers(0, 2, (B, Td, 256)).astype(np.uint8),
        'dec_keep2': rng.integers(0, 2, (B, Td, 128)).astype(np.uint8),
        'sample': rng.integers(0, 2, (Td, B)).astype(np.uint8),
    }
    inp = {'text': text, 'text_length': tl, 'mel': mel, 'stft': stft}
    fm = {k: v.astype(np.float64) for k, v in masks.items()}
    s2s, out, al, enc = on.forward(p, inp, r, Td, True, fm)
    loss = on.loss_fn(s2s, out, mel, stft)
    lt, s2, o2, a2, grads = ot.loss_and_grads(p, inp, r, Td, fm)
    assert abs(loss - lt) < 1e-8 * abs(loss) and np.abs(s2s - s2).max() < 1e-9 and np.abs(al - a2).max() < 1e-10
    is2s, iout, ial, _ = on.forward(p, inp, r, Td, False)

    def margins(a):
        srt = np.sort(a, -1)
        return srt[..., -1] - srt[..., -2]
    flat = on.flatten_params(p, V, r, np.float64)
    f32 = np.float32
    d = dict(
        r=r, V=V, B=B, Tt=Tt, Td=Td, seed=seed, perturb=0.3, param_checksum=float(np.abs(flat).sum()),
        scaled_names=np.array([k for k, _ in PEAKED_SCALES]), scaled_by=np.array([s for _, s in PEAKED_SCALES]),
        text=text, text_length=tl, mel=mel.astype(f32), stft=stft.astype(f32),
        seq2seq_output=s2s.astype(f32), output=out.astype(f32), alignments=al.astype(f32), loss=loss,
        argmax=al.argmax(-1).astype(np.int32), argmax_margin=margins(al),
        infer_seq2seq_output=is2s.astype(f32), infer_argmax=ial.argmax(-1).astype(np.int32), infer_argmax_margin=margins(ial),
        grad_names=np.array(list(grads.keys())),
        grad_norms=np.array([np.sqrt((g ** 2).sum()) for g in grads.values()]),
        assumptions=np.array(on.ASSUMPTIONS),
    )
    for k, v in masks.items():
        d['mask_' + k] = v
    mx = al.max(-1)
    print('peaked: max alpha median %.3f, > 0.9 on %.0f%% of (b,t), min margin %.2e, %d distinct arg-max positions' %
          (np.median(mx), 100 * (mx > 0.9).mean(), d['argmax_margin'].min(), len(np.unique(d['argmax']))))
    return d


def main():
    os.makedirs(OUT, exist_ok=True)
    for r, ns in ((2, 1), (5, 1), (2, 7)):
        d = make_case(r, num_speakers=ns)
        path = os.path.join(OUT, 'model_r%d%s.npz' % (r, '_spk' if ns > 1 else ''))
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path) // 1024, 'KiB', 'loss', d['loss'], 'min margin', d['argmax_margin'].min())
    path = os.path.join(OUT, 'model_r2_peaked.npz')
    np.savez_compressed(path, **make_peaked())
    print(path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
