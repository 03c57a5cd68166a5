"""CPU: the two independent restatements agree, gradients pass a finite-difference check, and the committed golden
vectors are reproduced.  (PARITY UNPINNED vs TensorFlow 1.2 -- see oracle/taco_numpy.py.)"""
import os

import numpy as np
import pytest
import torch

from oracle import taco_numpy as on
from oracle import taco_torch as ot
from tests.util import small_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _f64(masks):
    return {k: v.astype(np.float64) for k, v in masks.items()}


def test_param_count_matches_survey():
    # 6,926,609 trainable fp32 at V=60, r=2
    assert sum(int(np.prod(s)) for _, s, _ in on.param_spec(60, 2)) == 6926609


@pytest.mark.parametrize('r', [2, 5])
def test_numpy_and_torch_restatements_agree(r):
    V = 20
    p = on.init_params(V, r, seed=1, perturb=0.3)
    inp, masks = small_case(r=r, V=V, Td=4)
    inp = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in inp.items()}
    s2s, out, al, enc = on.forward(p, inp, r, 4, True, _f64(masks))
    loss = on.loss_fn(s2s, out, inp['mel'], inp['stft'])
    lt, s2, o2, a2, grads = ot.loss_and_grads(p, inp, r, 4, _f64(masks))
    assert abs(loss - lt) <= 1e-10 * abs(loss)
    assert np.abs(s2s - s2).max() < 1e-11 and np.abs(out - o2).max() < 1e-11 and np.abs(al - a2).max() < 1e-12
    assert all(g is not None for g in grads.values())
    # inference mode
    s2s, out, al, _ = on.forward(p, inp, r, 4, False)
    with torch.no_grad():
        t = ot.forward(ot.to_torch(p), {'text': torch.tensor(inp['text'], dtype=torch.int64),
                                        'text_length': torch.tensor(inp['text_length'], dtype=torch.int64)}, r, 4, False)
    assert np.abs(s2s - t[0].numpy()).max() < 1e-11 and np.abs(out - t[1].numpy()).max() < 1e-11


def test_multi_speaker_restatements_agree():
    """speaker table + encoder-CBHG speaker sites (ops.py:101-127)."""
    V, r, S = 20, 2, 5
    p = on.init_params(V, r, seed=1, perturb=0.3, num_speakers=S)
    inp, masks = small_case(r=r, V=V, Td=4)
    inp = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in inp.items()}
    inp['speaker'] = np.array([4, 1])
    s2s, out, al, _ = on.forward(p, inp, r, 4, True, _f64(masks))
    lt, s2, o2, a2, grads = ot.loss_and_grads(p, inp, r, 4, _f64(masks))
    assert np.abs(s2s - s2).max() < 1e-11 and np.abs(out - o2).max() < 1e-11 and np.abs(al - a2).max() < 1e-12
    assert np.linalg.norm(grads['speaker_embed']) > 0 and np.linalg.norm(grads['speaker_embed'][0]) == 0   # only used rows
    inp2 = dict(inp, speaker=np.array([2, 1]))
    assert np.abs(on.forward(p, inp2, r, 4, True, _f64(masks))[0] - s2s).max() > 1e-6   # the speaker id matters


def test_alignments_are_masked_distributions():
    p = on.init_params(20, 2, seed=2)
    inp, masks = small_case()
    _, _, al, _ = on.forward(p, {k: v for k, v in inp.items()}, 2, 5, False)
    assert np.allclose(al.sum(-1), 1.0)
    for b, L in enumerate(inp['text_length']):
        assert np.all(al[b, :, L:] == 0)


def test_autograd_matches_finite_differences():
    """Central differences of the NumPy forward vs torch autograd, on a few scalars of different tensors."""
    r, V, Td = 2, 12, 3
    p = on.init_params(V, r, seed=5, perturb=0.2)
    inp, masks = small_case(r=r, V=V, Tt=6, Td=Td, seed=11)
    inp = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in inp.items()}
    fm = _f64(masks)
    _, _, _, _, grads = ot.loss_and_grads(p, inp, r, Td, fm)

    def loss_of(pp):
        s2s, out, _, _ = on.forward(pp, inp, r, Td, True, fm)
        return on.loss_fn(s2s, out, inp['mel'], inp['stft'])

    rng = np.random.default_rng(0)
    for name in ['decoder/attention_v', 'decoder/gru_1/gates/kernel', 'encoder/cbhg/bank_3/kernel',
                 'decoder/memory_layer/kernel', 'post/cbhg/proj1_bn/gamma', 'decoder/pre_net/dense/kernel',
                 'encoder/cbhg/bigru/bw/candidate/kernel']:
        idx = tuple(rng.integers(0, s) for s in p[name].shape)
        eps = 1e-6
        base = p[name][idx]
        p[name][idx] = base + eps
        lp = loss_of(p)
        p[name][idx] = base - eps
        lm = loss_of(p)
        p[name][idx] = base
        fd = (lp - lm) / (2 * eps)
        an = grads[name][idx]
        # |x| kinks make the loss only piecewise smooth; tolerate small mismatch
        assert abs(fd - an) <= 2e-4 * max(1.0, abs(an)), (name, fd, an)


@pytest.mark.parametrize('name', ['model_r2', 'model_r5', 'model_r2_spk'])
def test_golden_fixture_reproduced(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    V, Td, r, S = int(g['V']), int(g['Td']), int(g['r']), int(g['num_speakers'])
    p = on.init_params(V, r, seed=int(g['seed']), perturb=float(g['perturb']), num_speakers=S)
    assert abs(np.abs(on.flatten_params(p, V, r, np.float64, S)).sum() - float(g['param_checksum'])) < 1e-9
    inp = {'text': g['text'], 'text_length': g['text_length'], 'mel': g['mel'].astype(np.float64),
           'stft': g['stft'].astype(np.float64)}
    if S > 1:
        inp['speaker'] = g['speaker']
    masks = {k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith('mask_')}
    s2s, out, al, enc = on.forward(p, inp, r, Td, True, masks)
    assert np.abs(s2s - g['seq2seq_output']).max() < 1e-12
    assert np.abs(out - g['output']).max() < 1e-12
    assert np.abs(al - g['alignments']).max() < 1e-13
    assert abs(on.loss_fn(s2s, out, inp['mel'], inp['stft']) - float(g['loss'])) < 1e-9
    assert len(g['assumptions']) == len(on.ASSUMPTIONS)


def test_peaked_fixture_reproduced():
    """tests/golden/model_r2_peaked.npz (peaked attention, stored as fp32) is what both restatements compute."""
    g = np.load(os.path.join(GOLD, 'model_r2_peaked.npz'))
    V, Td, r = int(g['V']), int(g['Td']), int(g['r'])
    p = on.init_params(V, r, seed=int(g['seed']), perturb=float(g['perturb']))
    for k, sc in zip(g['scaled_names'], g['scaled_by']):
        p[str(k)] = p[str(k)] * float(sc)
    assert abs(np.abs(on.flatten_params(p, V, r, np.float64)).sum() - float(g['param_checksum'])) < 1e-8
    inp = {'text': g['text'], 'text_length': g['text_length'], 'mel': g['mel'].astype(np.float64),
           'stft': g['stft'].astype(np.float64)}
    masks = {k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith('mask_')}
    s2s, out, al, _ = on.forward(p, inp, r, Td, True, masks)
    assert np.abs(s2s - g['seq2seq_output']).max() < 1e-5 and np.abs(out - g['output']).max() < 1e-5
    assert np.abs(al - g['alignments']).max() < 1e-7
    assert np.array_equal(al.argmax(-1), g['argmax'])
    assert float((al.max(-1) > 0.9).mean()) > 0.5 and g['argmax_margin'].min() > 1e-3
    lt, s2, o2, a2, _ = ot.loss_and_grads(p, inp, r, Td, masks)
    assert np.abs(a2 - al).max() < 1e-10 and abs(lt - float(g['loss'])) < 1e-8 * lt


def test_clip_adam_restatements_agree():
    rng = np.random.default_rng(0)
    p = {'a': rng.standard_normal((5, 3)), 'b': rng.standard_normal(7)}
    g = {'a': rng.standard_normal((5, 3)) * 10, 'b': rng.standard_normal(7) * 10}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(x) for k, x in p.items()}
    pt = {k: torch.tensor(x) for k, x in p.items()}
    gt = {k: torch.tensor(x) for k, x in g.items()}
    mt = {k: torch.zeros_like(x) for k, x in pt.items()}
    vt = {k: torch.zeros_like(x) for k, x in pt.items()}
    for step in (1, 2, 3):
        gn = on.clip_adam_step(p, g, m, v, step, 5e-4)
        gn2 = ot.clip_adam_step(pt, gt, mt, vt, step, 5e-4)
        assert abs(gn - gn2) < 1e-9
    for k in p:
        assert np.abs(p[k] - pt[k].numpy()).max() < 1e-12
    assert gn > 5  # the clip branch was exercised


def test_griffinlim_oracle_stft_istft_are_consistent():
    """oracle/griffinlim_numpy.py (librosa's algorithm restated; librosa itself is absent): istft inverts stft, stft agrees with
    scipy.signal.stft under matching conventions and with a hand-written DFT, Griffin-Lim reduces the spectral error."""
    import scipy.signal
    from oracle import griffinlim_numpy as gl
    rng = np.random.default_rng(0)
    y = rng.standard_normal(300 * 11)
    S = gl.stft(y)
    assert S.shape == (1025, 12)
    back = gl.istft(S)
    assert back.shape == y.shape and np.abs(back - y).max() < 1e-10          # exact inverse (window sum-of-squares normalisation)
    # independent implementation: scipy's STFT with the same window image, hop, reflect ('even') boundary, no scaling
    w = gl.pad_center(gl.hann_periodic())
    _, _, Z = scipy.signal.stft(y, window=w, nperseg=2048, noverlap=2048 - 300, nfft=2048, boundary='even', padded=False,
                                return_onesided=True, scaling='spectrum')
    Z = Z * w.sum()                                                          # undo scipy's 'spectrum' scaling
    assert Z.shape[1] >= 12 and np.abs(Z[:, :12] - S).max() < 1e-9 * np.abs(S).max()
    # one bin by hand
    yp = np.pad(y, 1024, mode='reflect')
    k, t = 37, 5
    n = np.arange(2048)
    assert abs(np.sum(w * yp[t * 300:t * 300 + 2048] * np.exp(-2j * np.pi * k * n / 2048)) - S[k, t]) < 1e-9
    mag = np.abs(S)
    ph = 2 * np.pi * rng.random(mag.shape)
    e0 = gl.spectral_convergence(gl.griffinlim(mag, ph, 0), mag)
    e20 = gl.spectral_convergence(gl.griffinlim(mag, ph, 20), mag)
    assert e20 < 0.6 * e0


def test_decisions_record_and_force_are_consistent():
    """oracle.taco_torch.Decisions: recording changes nothing, and forcing the graph's OWN decisions reproduces its loss and
    gradients (to summation-order rounding); forcing ONE flipped ReLU decision changes the gradient below that unit."""
    from tests.util import small_case
    r, V, B, Tt, Td = 2, 20, 2, 9, 5
    p = on.init_params(V, r, seed=3, perturb=0.2)
    inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td)
    f = lambda d: {k: (np.asarray(v, dtype=np.float64) if np.asarray(v).dtype.kind in 'fu' and k not in ('text', 'text_length') else v)
                   for k, v in d.items()}
    l0, _, _, _, g0 = ot.loss_and_grads(p, f(inp), r, Td, f(masks))
    dec = ot.Decisions()
    l1, _, _, _, g1 = ot.loss_and_grads(p, f(inp), r, Td, f(masks), dec=dec)
    assert l0 == l1
    sites = set(dec.rec)
    assert {'encoder/cbhg/bank', 'encoder/cbhg/pool', 'encoder/cbhg/proj1', 'encoder/cbhg/highway_3/H', 'post/cbhg/pool',
            'encoder/pre_net/l1', 'decoder/pre_net/l2@%d' % (Td - 1)} <= sites
    force = {k: v.to(torch.float64) for k, v in dec.rec.items()}
    l2, _, _, _, g2 = ot.loss_and_grads(p, f(inp), r, Td, f(masks), dec=ot.Decisions(force))
    assert abs(l2 - l0) <= 1e-12 * abs(l0)
    for k in g0:
        if g0[k] is not None:
            assert np.abs(g0[k] - g1[k]).max() <= 1e-12 * (np.abs(g0[k]).max() + 1e-300), k
            assert np.abs(g0[k] - g2[k]).max() <= 1e-12 * (np.abs(g0[k]).max() + 1e-300), k
    flipped = dict(force)
    m = flipped['encoder/cbhg/bank'].clone()
    m[0, 4, 7] = 1.0 - m[0, 4, 7]
    flipped['encoder/cbhg/bank'] = m
    _, _, _, _, g3 = ot.loss_and_grads(p, f(inp), r, Td, f(masks), dec=ot.Decisions(flipped))
    assert np.abs(g3['encoder/pre_net/dense_1/kernel'] - g0['encoder/pre_net/dense_1/kernel']).max() > 0
