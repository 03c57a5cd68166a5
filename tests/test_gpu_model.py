"""GPU: whole-path parity of libtaco_hip.so against the CPU restatements (PARITY UNPINNED vs TF 1.2, see oracle/).

Stated tolerances (fp32 HIP vs fp64 oracle): seq2seq_output / output rel-L2 <= 1e-5 and max-abs <= 5e-5; alignments
max-abs <= 1e-6; loss rel <= 1e-5; per-tensor gradients rel-L2 <= 2e-4 (tensors whose reference norm is < 1e-6 of the
largest are compared in absolute terms); attention argmax exact wherever the reference top-1/top-2 margin >= 1e-4.
(Measured on MI355X in round 1: outputs ~3e-7 rel-L2, worst gradient 1.1e-5.)
"""
import os

import numpy as np
import pytest
import torch

from oracle import taco_numpy as on
from oracle import taco_torch as ot
from tests.util import max_abs, rel_l2, report, small_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Runner:
    """Thin harness over the C ABI (no tacotron_amd.model involved, so op/host bugs separate cleanly)."""

    def __init__(self, lib, B, Tt, Td, r, V, train=True, S=1):
        from tacotron_amd.params import ParamBuffer
        self.lib, self.train = lib, train
        self.speaker = None
        self.shape = lib.make_shape(B, Tt, Td, r, V, S)
        self.pb = ParamBuffer(self.shape, 'cuda')
        self.ws = torch.zeros(lib.workspace_bytes(self.shape, train) // 4, device='cuda')
        self.s2s = torch.zeros(B, Td, 80 * r, device='cuda')
        self.out = torch.zeros(B, Td, 1025 * r, device='cuda')
        self.al = torch.zeros(B, Td, Tt, device='cuda')
        self.loss = torch.zeros(3, device='cuda')
        self.grads = torch.zeros(self.pb.numel, device='cuda')
        self.wtab = {n: (o, s, d) for n, o, s, d in lib.workspace_table(self.shape, train)}

    def set(self, p, inp, masks=None):
        self.pb.load_dict_(p)
        self.text = torch.as_tensor(inp['text']).to('cuda', torch.int32).contiguous()
        self.tl = torch.as_tensor(inp['text_length']).to('cuda', torch.int32).contiguous()
        if 'speaker' in inp:
            self.speaker = torch.as_tensor(inp['speaker']).to('cuda', torch.int32).contiguous()
        if 'mel' in inp:
            self.mel = torch.as_tensor(inp['mel']).to('cuda', torch.float32).contiguous()
            self.stft = torch.as_tensor(inp['stft']).to('cuda', torch.float32).contiguous()
        self.masks = {k: torch.as_tensor(v).to('cuda', torch.uint8).contiguous() for k, v in (masks or {}).items()}

    def forward(self):
        self.lib.forward(self.shape, self.pb.flat, self.text, self.tl, self.mel, self.stft, self.masks, self.s2s, self.out,
                         self.al, self.loss, self.ws, self.speaker)
        torch.cuda.synchronize()
        self.check_err()

    def backward(self):
        self.lib.backward(self.shape, self.pb.flat, self.text, self.tl, self.s2s, self.al, self.masks, self.grads, self.ws,
                          self.speaker)
        torch.cuda.synchronize()
        self.check_err()

    def infer(self):
        self.lib.infer(self.shape, self.pb.flat, self.text, self.tl, self.s2s, self.out, self.al, self.ws, self.speaker)
        torch.cuda.synchronize()
        self.check_err()

    def check_err(self):
        o, s, d = self.wtab['dec.err']
        flags = self.ws[o:o + 2].view(torch.int32).cpu().numpy()
        assert flags[0] == 0 and flags[1] == 0, 'decoder cluster exchange timed out: %s' % flags

    def wsget(self, name):
        o, s, d = self.wtab[name]
        return self.ws[o:o + s].view(*d).cpu().numpy()


def f64(d):
    return {k: (np.asarray(v, dtype=np.float64) if np.asarray(v).dtype.kind == 'f' or np.asarray(v).dtype == np.uint8 else v)
            for k, v in d.items()}


def golden(r, spk=False):
    g = np.load(os.path.join(GOLD, 'model_r%d%s.npz' % (r, '_spk' if spk else '')))
    V, S = int(g['V']), int(g['num_speakers'])
    p = on.init_params(V, r, seed=int(g['seed']), perturb=float(g['perturb']), num_speakers=S)
    inp = {'text': g['text'], 'text_length': g['text_length'], 'mel': g['mel'], 'stft': g['stft']}
    if S > 1:
        inp['speaker'] = g['speaker']
    masks = {k[5:]: g[k] for k in g.files if k.startswith('mask_')}
    return g, p, inp, masks


@pytest.mark.parametrize('r', [2, 5])
def test_encoder_stages_vs_oracle(built_lib, r):
    """Stage-by-stage: localises a divergence to one kernel family."""
    g, p, inp, masks = golden(r)
    B, Tt, Td, V = int(g['B']), int(g['Tt']), int(g['Td']), int(g['V'])
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    R.forward()
    k1, k2 = masks['enc_keep1'].astype(np.float64), masks['enc_keep2'].astype(np.float64)
    emb = p['embedding'][inp['text']]
    pre = on.pre_net(emb, p, 'encoder/pre_net/', k1, k2)
    M1 = B * Tt
    assert report('enc.emb', R.wsget('enc.emb'), emb.reshape(M1, -1))[0] < 1e-6
    assert report('enc.p2', R.wsget('enc.p2'), pre.reshape(M1, -1))[0] < 1e-5
    pf = 'encoder/cbhg/'
    bank = np.concatenate([on.relu(on.conv1d_same(pre, p[pf + 'bank_%d/kernel' % k], p[pf + 'bank_%d/bias' % k]))
                           for k in range(1, 17)], -1)
    assert report('enc.bank', R.wsget('enc.bank'), bank.reshape(M1, -1))[0] < 1e-5
    pool = on.maxpool2_same(on.bn_affine(bank, p[pf + 'bank_bn/gamma'], p[pf + 'bank_bn/beta']))
    assert report('enc.pool', R.wsget('enc.pool'), pool.reshape(M1, -1))[0] < 1e-5
    y = on.relu(on.conv1d_same(pool, p[pf + 'proj1/kernel'], p[pf + 'proj1/bias']))
    assert report('enc.pj1pre', R.wsget('enc.pj1pre'), y.reshape(M1, -1))[0] < 1e-5
    y = on.bn_affine(y, p[pf + 'proj1_bn/gamma'], p[pf + 'proj1_bn/beta'])
    assert report('enc.pj1', R.wsget('enc.pj1'), y.reshape(M1, -1))[0] < 1e-5
    y = on.bn_affine(on.conv1d_same(y, p[pf + 'proj2/kernel'], p[pf + 'proj2/bias']), p[pf + 'proj2_bn/gamma'],
                     p[pf + 'proj2_bn/beta'])
    h = y + pre
    assert report('enc.res', R.wsget('enc.res'), h.reshape(M1, -1))[0] < 1e-5
    for l in range(4):
        h = on.highway(h, p, pf + 'highway_%d/' % l)
        assert report('enc.h%d' % (l + 1), R.wsget('enc.h%d' % (l + 1)), h.reshape(M1, -1))[0] < 2e-5
    enc = on.bigru(h, p, pf + 'bigru/')
    assert report('enc.out', R.wsget('enc.out'), enc.reshape(M1, -1))[0] < 3e-5
    values, keys, _ = on.attention_memory(p, enc, inp['text_length'])
    assert report('dec.values', R.wsget('dec.values'), values.reshape(M1, -1))[0] < 3e-5
    assert report('dec.keys', R.wsget('dec.keys'), keys.reshape(M1, -1))[0] < 3e-5


@pytest.mark.parametrize('r', [2, 5])
def test_forward_train_matches_golden(built_lib, r):
    g, p, inp, masks = golden(r)
    R = Runner(built_lib, int(g['B']), int(g['Tt']), int(g['Td']), r, int(g['V']))
    R.set(p, inp, masks)
    R.forward()
    s2s, out, al = R.s2s.cpu().numpy(), R.out.cpu().numpy(), R.al.cpu().numpy()
    r1, m1 = report('seq2seq_output', s2s, g['seq2seq_output'])
    r2, m2 = report('output', out, g['output'])
    r3, m3 = report('alignments', al, g['alignments'])
    loss = R.loss.cpu().numpy()
    print('  loss', loss, float(g['loss']))
    assert r1 < 1e-5 and m1 < 5e-5 and r2 < 1e-5 and m2 < 5e-5 and m3 < 1e-6
    assert abs(loss[0] - float(g['loss'])) <= 1e-5 * float(g['loss'])
    assert abs(loss[0] - (loss[1] + loss[2])) <= 1e-5 * loss[0]
    ok = g['argmax_margin'] >= 1e-4
    assert np.array_equal(al.argmax(-1)[ok], g['alignments'].argmax(-1)[ok]), 'attention argmax differs'
    print('  argmax compared on %d/%d (b,t); min margin %.2e' % (ok.sum(), ok.size, g['argmax_margin'].min()))
    for b, L in enumerate(inp['text_length']):
        assert np.all(al[b, :, L:] == 0)


@pytest.mark.parametrize('r', [2, 5])
def test_infer_matches_golden(built_lib, r):
    g, p, inp, _ = golden(r)
    R = Runner(built_lib, int(g['B']), int(g['Tt']), int(g['Td']), r, int(g['V']), train=False)
    R.set(p, {'text': inp['text'], 'text_length': inp['text_length']})
    R.infer()
    r1, m1 = report('infer seq2seq_output', R.s2s.cpu().numpy(), g['infer_seq2seq_output'])
    r2, m2 = report('infer output', R.out.cpu().numpy(), g['infer_output'])
    r3, m3 = report('infer alignments', R.al.cpu().numpy(), g['infer_alignments'])
    assert r1 < 1e-5 and m1 < 5e-5 and r2 < 1e-5 and m2 < 5e-5 and m3 < 1e-6


@pytest.mark.parametrize('case', ['keys-beyond-bound', 'queries-beyond-bound', 'mixed', 'large-opposite-below-bound',
                                  'large-opposite-around-bound'])
def test_attention_scores_beyond_the_product_form_bound(built_lib, case):
    """decoder3's kernels form tanh(keys + q) (forward: the energies; BPTT: the energy backward) from a product of exponentials
    exp(2 k) exp(2 q) while |k|, |q| <= 8 and fall back, wave by wave and step by step, to the exact sum form beyond that
    (decoder3.hip kTanhSplit / kTanhSplitB).  Forced here: the memory
    layer scaled until most keys exceed the bound (every wave on the exact form), the query layer scaled until the queries do
    (the per-step fallback), and a scale at which only part of the keys do (both forms inside one launch).  Same tolerances as
    the unscaled fixture: the fallback must be the reference computation, not an approximation of it; no NaN from inf * 0.
    The two `large-opposite` cases are the PRECISION side of the bound: keys and queries of magnitude 4-8 (resp. 4-40)
    whose sum is near 0 -- the product form rounds each exponent at |k| and |q|, the sum form at |k + q| -- must still meet the
    unscaled fixture's tolerances; the share of such (key, query) pairs in the case is printed and asserted to be non-trivial."""
    g, p, inp, masks = golden(2)
    p = dict(p)
    mem = [k for k in p if k.endswith('memory_layer/kernel')][0]
    qry = [k for k in p if k.endswith('query_layer/kernel')][0]
    if case == 'keys-beyond-bound':
        p[mem] = p[mem] * 400.0
    elif case == 'queries-beyond-bound':
        p[qry] = p[qry] * 3000.0
    elif case == 'mixed':
        p[mem] = p[mem] * 60.0
        p[qry] = p[qry] * 300.0
    elif case == 'large-opposite-below-bound':    # |keys| <= 6.9, |q| <= 7.5: every element on the product form
        p[mem] = p[mem] * 15.0
        p[qry] = p[qry] * 3.5
    else:                                          # |keys| <= 27.5, |q| <= 25.5, a third of each beyond the bound: both forms
        p[mem] = p[mem] * 60.0
        p[qry] = p[qry] * 12.0
    B, Tt, Td, V = int(g['B']), int(g['Tt']), int(g['Td']), int(g['V'])
    p64 = f64(p)
    s2s, out, al, extra = on.forward(p64, f64(inp), 2, Td, True, {k: v.astype(np.float64) for k, v in masks.items()})
    if case.startswith('large-opposite'):
        # which (key, query) pairs does this case hold?  keys (B, Tt, 256) from the oracle's encoder, queries (B, Td, 256) from its outputs
        keys = on.attention_memory(p64, extra, inp['text_length'])[1]
        q = s2s @ p64[qry]
        kk, qq = keys[:, None, :, :], q[:, :, None, :]
        pairs = (np.abs(kk) > 3.0) & (np.abs(qq) > 3.0) & (np.abs(kk + qq) < 1.0)
        print('  %s: |keys| max %.1f, |q| max %.1f; %d of %d (key, query, unit) triples have |k|, |q| > 3 and |k + q| < 1'
              % (case, np.abs(keys).max(), np.abs(q).max(), int(pairs.sum()), pairs.size))
        assert pairs.sum() >= 100
    R = Runner(built_lib, B, Tt, Td, 2, V)
    R.set(p, inp, masks)
    R.forward()
    a_hip = R.al.cpu().numpy()   # (Runner.forward has asserted that both decoder error words are clear)
    assert np.isfinite(a_hip).all() and np.isfinite(R.s2s.cpu().numpy()).all()
    r1, m1 = report('%s seq2seq_output' % case, R.s2s.cpu().numpy(), s2s)
    r3, m3 = report('%s alignments' % case, a_hip, al)
    assert r1 < 1e-5 and m1 < 5e-5 and m3 < 2e-6
    # ... and the BPTT kernel's energy backward, which takes the same decision per wave and step
    R.backward()
    ref = ot.loss_and_grads(p64, inp, 2, Td, {k: v.astype(np.float64) for k, v in masks.items()})
    assert abs(float(R.loss[0]) - ref[0]) <= 1e-5 * abs(ref[0])
    # (with a saturated tanh the attention parameters' gradients are sums that cancel analytically -- d v_u = +-sum_s de_s = +-0 --
    #  so their norms are ~1e-4 of the model's and fp32 resolves them to ~7e-4 of THEMSELVES whatever the formula; every tensor is
    #  therefore held to 2e-4 of max(its own norm, 1e-3 of the largest gradient norm))
    got = R.pb.to_dict(R.grads)
    gmax = max(np.linalg.norm(v) for v in ref[4].values() if v is not None)
    for name, gr in ref[4].items():
        gr = np.zeros_like(got[name]) if gr is None else gr
        err = np.linalg.norm(got[name] - gr) / max(np.linalg.norm(gr), 1e-3 * gmax)
        assert np.isfinite(got[name]).all() and err <= 2e-4, (name, err)
    Ri = Runner(built_lib, B, Tt, Td, 2, V, train=False)
    Ri.set(p, {'text': inp['text'], 'text_length': inp['text_length']})
    Ri.infer()
    s2i, outi, ali = on.forward(p64, f64(inp), 2, Td, train=False, masks=None)[:3]
    assert report('%s infer alignments' % case, Ri.al.cpu().numpy(), ali)[1] < 2e-6
    assert report('%s infer seq2seq_output' % case, Ri.s2s.cpu().numpy(), s2i)[0] < 1e-5


def check_grads(R, ref_grads, tol=2e-4):
    got = R.pb.to_dict(R.grads)
    gmax = max(np.linalg.norm(v) for v in ref_grads.values() if v is not None)
    bad = []
    for name, ref in ref_grads.items():
        if ref is None:   # autograd: the parameter did not take part (e.g. the attention layer when Td == 1)
            ref = np.zeros_like(got[name])
        nr = np.linalg.norm(ref)
        if nr < 1e-6 * gmax:
            err = np.linalg.norm(got[name] - ref) / gmax
        else:
            err = rel_l2(got[name], ref)
        flag = '' if err < tol else '   <-- FAIL'
        print('  grad %-45s |ref|=%.3e err=%.3e%s' % (name, nr, err, flag))
        if err >= tol:
            bad.append((name, err))
    return bad


@pytest.mark.parametrize('r', [2, 5])
def test_backward_matches_autograd_golden(built_lib, r):
    g, p, inp, masks = golden(r)
    R = Runner(built_lib, int(g['B']), int(g['Tt']), int(g['Td']), r, int(g['V']))
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    _, _, _, _, ref = ot.loss_and_grads(p, f64(inp), r, int(g['Td']), f64(masks))
    # the committed per-tensor norms pin the autograd reference itself
    names = [str(n) for n in g['grad_names']]
    for n, gn in zip(names, g['grad_norms']):
        assert abs(np.linalg.norm(ref[n]) - gn) <= 1e-9 * max(1.0, gn)
    bad = check_grads(R, ref)
    assert not bad, bad


@pytest.mark.parametrize('form', ['fused', 'per-layer'])
def test_multi_speaker_forward_backward_infer(built_lib, form, monkeypatch):
    """SURVEY §8 a4/a9 / BASELINE config 5: speaker table + encoder CBHG speaker sites, against the committed fixture
    and torch autograd.  `fused` (default since round 6): the four highway layers with their adapters as one launch per direction
    and the speaker sites' small chains in grouped launches; `per-layer` (TACO_SPK_UNFUSED=1): the launches of rounds 1-5."""
    if form == 'per-layer':
        monkeypatch.setenv('TACO_SPK_UNFUSED', '1')
    g, p, inp, masks = golden(2, spk=True)
    B, Tt, Td, V, S = int(g['B']), int(g['Tt']), int(g['Td']), int(g['V']), int(g['num_speakers'])
    R = Runner(built_lib, B, Tt, Td, 2, V, S=S)
    R.set(p, inp, masks)
    R.forward()
    r1, m1 = report('spk seq2seq_output', R.s2s.cpu().numpy(), g['seq2seq_output'])
    r2, m2 = report('spk output', R.out.cpu().numpy(), g['output'])
    r3, m3 = report('spk alignments', R.al.cpu().numpy(), g['alignments'])
    assert report('spk enc.out', R.wsget('enc.out'), g['encoded'].reshape(B * Tt, -1))[0] < 3e-5
    assert r1 < 1e-5 and m1 < 5e-5 and r2 < 1e-5 and m2 < 5e-5 and m3 < 1e-6
    assert abs(R.loss[0].item() - float(g['loss'])) <= 1e-5 * float(g['loss'])
    R.backward()
    _, _, _, _, ref = ot.loss_and_grads(p, f64(inp), 2, Td, f64(masks))
    bad = check_grads(R, ref)
    assert not bad, bad
    Ri = Runner(built_lib, B, Tt, Td, 2, V, train=False, S=S)
    Ri.set(p, {'text': inp['text'], 'text_length': inp['text_length'], 'speaker': inp['speaker']})
    Ri.infer()
    assert report('spk infer output', Ri.out.cpu().numpy(), g['infer_output'])[0] < 1e-5
    # a model built for speakers refuses to run without ids (loud failure, not a silent single-speaker fallback)
    Ri.speaker = None
    with pytest.raises(built_lib.TacoError):
        Ri.infer()


def test_backward_without_masks_and_ragged_lengths(built_lib):
    """No dropout, no scheduled sampling (TrainingHelper path, tacotron.py:86-87), odd sizes, short rows."""
    r, V, B, Tt, Td = 2, 17, 3, 13, 7
    p = on.init_params(V, r, seed=9, perturb=0.3)
    inp, _ = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=21, full_len_row0=False)
    inp['text_length'][:] = [1, 13, 6]
    inp['text'][0, 1:] = 0
    inp['text'][2, 6:] = 0
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, None)
    R.forward()
    R.backward()
    lt, s2, o2, a2, ref = ot.loss_and_grads(p, f64(inp), r, Td, None)
    assert report('s2s', R.s2s.cpu().numpy(), s2)[0] < 1e-5
    assert report('out', R.out.cpu().numpy(), o2)[0] < 1e-5
    assert report('align', R.al.cpu().numpy(), a2)[1] < 1e-6
    assert abs(R.loss[0].item() - lt) <= 1e-5 * lt
    bad = check_grads(R, ref)
    assert not bad, bad


@pytest.mark.parametrize('Td', [1, 2])
def test_shortest_decodes(built_lib, Td):
    """Td = 1 (no next step anywhere: every deferred / prefetched piece of the folded decoder rounds is skipped) and
    Td = 2, with dropout and scheduled sampling, forward + backward + inference."""
    r, V, B, Tt = 2, 19, 2, 9
    p = on.init_params(V, r, seed=4, perturb=0.3)
    inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=33)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    lt, s2, o2, a2, ref = ot.loss_and_grads(p, f64(inp), r, Td, f64(masks))
    assert report('s2s', R.s2s.cpu().numpy(), s2)[0] < 1e-5
    assert report('out', R.out.cpu().numpy(), o2)[0] < 1e-5
    assert report('align', R.al.cpu().numpy(), a2)[1] < 1e-6
    bad = check_grads(R, ref)
    assert not bad, bad
    Ri = Runner(built_lib, B, Tt, Td, r, V, train=False)
    Ri.set(p, {'text': inp['text'], 'text_length': inp['text_length']})
    Ri.infer()
    si, oi, ai = on.forward(p, f64(inp), r, Td, train=False, masks=None)[:3]
    assert report('infer s2s', Ri.s2s.cpu().numpy(), si)[0] < 1e-5
    assert report('infer align', Ri.al.cpu().numpy(), ai)[1] < 1e-6


@pytest.mark.parametrize('cluster', ['32', '16', '8', '4', '2', '1'])
def test_long_text_and_cluster_widths(built_lib, cluster, monkeypatch):
    """Tt = 300 (> 256) leaves the register-resident attention rows and exercises the streamed fall-back paths of both
    decoder kernels; TACO_DEC_CLUSTER forces the narrower cluster widths used when B * 8 workgroups are not co-resident;
    B = 1 is the single-prompt inference shape."""
    monkeypatch.setenv('TACO_DEC_CLUSTER', cluster)
    r, V, B, Tt, Td = 2, 25, 2, 300, 4
    p = on.init_params(V, r, seed=6, perturb=0.2)
    inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=12)
    inp['text_length'][:] = [300, 211]
    inp['text'][1, 211:] = 0
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    lt, s2, o2, a2, ref = ot.loss_and_grads(p, f64(inp), r, Td, f64(masks))
    assert report('s2s (Tt=300, P=%s)' % cluster, R.s2s.cpu().numpy(), s2)[0] < 1e-5
    assert report('out', R.out.cpu().numpy(), o2)[0] < 1e-5
    assert report('align', R.al.cpu().numpy(), a2)[1] < 1e-6
    bad = check_grads(R, ref)
    assert not bad, bad
    # single-row inference
    R1 = Runner(built_lib, 1, Tt, Td, r, V, train=False)
    R1.set(p, {'text': inp['text'][:1], 'text_length': inp['text_length'][:1]})
    R1.infer()
    s1, o1, a1, _ = on.forward({k: v for k, v in p.items()}, {'text': inp['text'][:1], 'text_length': inp['text_length'][:1]},
                               r, Td, False)
    assert report('B=1 infer out', R1.out.cpu().numpy(), o1)[0] < 1e-5
    assert report('B=1 infer align', R1.al.cpu().numpy(), a1)[1] < 1e-6


def test_medium_shape_forward_backward(built_lib, grad_tol=2e-4):
    """B=4, Tt=37, Td=12: multi-tile GEMMs, several attention rows per wave, sampling + dropout masks."""
    r, V, B, Tt, Td = 2, 40, 4, 37, 12
    p = on.init_params(V, r, seed=4, perturb=0.2)
    inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=8)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    adj, _ = l1_tie_adjusted(R, p, inp, masks, r, Td)
    lt, s2, o2, a2, ref = ot.loss_and_grads(p, f64(adj), r, Td, f64(masks))
    assert report('s2s', R.s2s.cpu().numpy(), s2)[0] < 1e-5
    assert report('out', R.out.cpu().numpy(), o2)[0] < 1e-5
    assert report('align', R.al.cpu().numpy(), a2)[1] < 1e-6
    assert abs(R.loss[0].item() - lt) <= 1e-5 * lt
    bad = check_grads(R, ref, tol=grad_tol)
    assert not bad, bad


def test_full_size_properties(built_lib):
    """BASELINE S1 shape (B=32, Tt=200, Td=180, r=2): size-independent properties instead of a CPU comparison."""
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.params import ParamBuffer
    B, Tt, Td, r, V = 32, 200, 180, 2, 60
    batch = synthetic_batch(B, Tt, Td, r, V)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.pb.init_(seed=0)
    p = R.pb.to_dict()
    inp = {k: batch[k].numpy() for k in ('text', 'text_length', 'mel', 'stft')}
    rng = np.random.default_rng(0)
    masks = {'enc_keep1': rng.integers(0, 2, (B, Tt, 256)), 'enc_keep2': rng.integers(0, 2, (B, Tt, 128)),
             'dec_keep1': rng.integers(0, 2, (B, Td, 256)), 'dec_keep2': rng.integers(0, 2, (B, Td, 128)),
             'sample': rng.integers(0, 2, (Td, B))}
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    s2s, out, al = R.s2s, R.out, R.al
    assert torch.isfinite(s2s).all() and torch.isfinite(out).all() and torch.isfinite(R.grads).all()
    assert float((al.sum(-1) - 1).abs().max()) < 1e-5
    for b, L in enumerate(inp['text_length']):
        assert float(al[b, :, L:].abs().max()) == 0 if L < Tt else True
    # loss == recomputed L1 sums (independent torch reduction in fp64)
    l1 = (s2s.double() - R.mel.double()).abs().sum().item()
    l2 = (out.double() - R.stft.double()).abs().sum().item()
    loss = R.loss.cpu().numpy()
    print('  loss', loss, l1, l2)
    assert abs(loss[1] - l1) <= 2e-5 * l1 and abs(loss[2] - l2) <= 2e-5 * l2
    # determinism of the forward path
    s2s0, out0 = s2s.clone(), out.clone()
    g_full = R.grads.clone()
    R.forward()
    assert torch.equal(s2s0, R.s2s) and torch.equal(out0, R.out)
    # data-parallel additivity (loss is a SUM): grads(batch) == grads(first half) + grads(second half)
    halves = []
    for sl in (slice(0, 16), slice(16, 32)):
        Rh = Runner(built_lib, 16, Tt, Td, r, V)
        mh = {k: (v[:, sl] if k == 'sample' else v[sl]) for k, v in masks.items()}
        Rh.set(p, {k: v[sl] for k, v in inp.items()}, mh)
        Rh.forward()
        Rh.backward()
        # rows are independent.  (Round 1 pinned this bitwise; since round 2 the GEMM dispatcher picks kernels and k-split
        # factors by tile count, i.e. by batch size, so the summation order -- not the result -- may differ between B = 16 and 32.)
        dh = float((Rh.s2s - s2s0[sl]).abs().max())
        print('  rows of a B=16 run vs the same rows of the B=32 run: max|d s2s| = %.2e (bitwise: %s)' % (dh, dh == 0.0))
        assert dh <= 2e-6 * float(s2s0.abs().max())
        halves.append(Rh.grads.clone())
        del Rh
    gsum = halves[0] + halves[1]
    err = (gsum - g_full).norm().item() / g_full.norm().item()
    print('  DP additivity rel err %.3e, |g|=%.4e' % (err, g_full.norm().item()))
    assert err < 1e-5


@pytest.mark.parametrize('variant', ['32x2', '16x3'])
def test_medium_shape_with_gemm2_forced(built_lib, variant, monkeypatch):
    """The whole train step with EVERY eligible NN GEMM on the second-generation kernel (at the test sizes the dispatcher would
    otherwise keep the 64x64 kernel): batched conv-bank launch, tap-split projections, atomic-accumulate bank backward."""
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_VARIANT', variant)
    # Same 2e-4 gradient tolerance as everywhere else.  (Round 2 ran this at 1e-3 because the encoder pre_net / embedding
    # gradients once moved by 2.3e-4 with this kernel's summation order; the cause is the L1 loss's sign() at a rounding-level
    # tie -- one flipped sign in d loss / d output -- which l1_tie_adjusted now takes out of the comparison.)
    test_medium_shape_forward_backward(built_lib, grad_tol=2e-4)
    test_backward_without_masks_and_ragged_lengths(built_lib)


@pytest.mark.parametrize('B,Tt,Td', [(3, 127, 64), (2, 128, 9), (1, 255, 12), (4, 200, 100), (5, 9, 5), (2, 254, 127)])
def test_pooled_bank_epilogue_tile_edges(built_lib, B, Tt, Td, monkeypatch):
    """Conv bank + BN-affine + max-pool as ONE launch (gemm2.hip pooled epilogue, ops.py:54-71): sequence lengths chosen so that
    sequence ends fall on every special row of the overlapping 128-row tiles (last row of a tile = halo, first row of the next,
    rows 3|4 / 7|8 / 31|32 of the MFMA layout), for both CBHGs (T = Tt and T = Td * r).  Checked against the CPU restatement
    and -- bitwise -- against the two-pass form (TACO_NO_POOL_FUSE=1) on the same inputs."""
    r, V = 2, 40
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    p = on.init_params(V, r, seed=11, perturb=0.3)
    inp, masks = _full_case(B, Tt, Td, r, V)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('TACO_NO_POOL_FUSE', mode)
        R.forward()
        got[mode] = {k: R.wsget(k).copy() for k in ('enc.bank', 'enc.pool', 'post.bank', 'post.pool', 'enc.p2')}
        got[mode]['s2s'] = R.s2s.cpu().numpy().copy()
        # ... and the backward twin (the max-pool / BN / ReLU backward as the epilogue of the d pool GEMM, gemm2.hip pool == 2)
        R.backward()
        got[mode]['grads'] = R.grads.cpu().numpy().copy()
    ga, gb = got['0']['grads'].astype(np.float64), got['1']['grads'].astype(np.float64)
    rel = np.linalg.norm(ga - gb) / np.linalg.norm(gb)
    print('  gradients, fused vs two-pass bank backward: rel-L2 %.2e, max|d| %.2e (|g|max %.2e)' % (rel, np.abs(ga - gb).max(), np.abs(gb).max()))
    assert rel < 2e-6          # same arithmetic; only the order of the atomic accumulations differs
    for name in ('encoder/cbhg/bank_bn/gamma', 'encoder/cbhg/bank_bn/beta', 'post/cbhg/bank_bn/gamma', 'post/cbhg/bank_bn/beta',
                 'encoder/cbhg/bank_16/kernel', 'post/cbhg/bank_1/kernel', 'embedding'):
        o, n_ = R.pb.index[name][0], R.pb.index[name][1]
        d = np.linalg.norm(ga[o:o + n_] - gb[o:o + n_]) / max(1e-30, np.linalg.norm(gb[o:o + n_]))
        assert d < 1e-5, (name, d)
    for k in ('enc.bank', 'enc.pool', 'post.bank', 'post.pool'):
        assert np.array_equal(got['0'][k], got['1'][k]), k
    for pf, xin, T, K in (('encoder/cbhg/', got['0']['enc.p2'].reshape(B, Tt, -1).astype(np.float64), Tt, 16),
                          ('post/cbhg/', got['0']['s2s'].reshape(B, Td * r, 80).astype(np.float64), Td * r, 8)):
        bank = np.concatenate([on.relu(on.conv1d_same(xin, p[pf + 'bank_%d/kernel' % k], p[pf + 'bank_%d/bias' % k]))
                               for k in range(1, K + 1)], -1)
        pool = on.maxpool2_same(on.bn_affine(bank, p[pf + 'bank_bn/gamma'], p[pf + 'bank_bn/beta']))
        tag = 'enc' if K == 16 else 'post'
        assert report(tag + '.bank', got['0'][tag + '.bank'], bank.reshape(B * T, -1))[0] < 1e-5
        assert report(tag + '.pool', got['0'][tag + '.pool'], pool.reshape(B * T, -1))[0] < 1e-5
    # inference keeps no un-pooled activations: same pooled tensor
    Ri = Runner(built_lib, B, Tt, Td, r, V, train=False)
    Ri.set(p, inp)
    Ri.infer()
    xin = Ri.wsget('enc.p2').reshape(B, Tt, -1).astype(np.float64)       # (no dropout at inference: its own pre_net output)
    pf = 'encoder/cbhg/'
    bank = np.concatenate([on.relu(on.conv1d_same(xin, p[pf + 'bank_%d/kernel' % k], p[pf + 'bank_%d/bias' % k]))
                           for k in range(1, 17)], -1)
    pool = on.maxpool2_same(on.bn_affine(bank, p[pf + 'bank_bn/gamma'], p[pf + 'bank_bn/beta']))
    assert report('enc.pool (inference)', Ri.wsget('enc.pool'), pool.reshape(B * Tt, -1))[0] < 1e-5


@pytest.mark.parametrize('knob', ['TACO_NO_BANK_GATHER=1', 'TACO_GEMM2_XCD=0', 'TACO_GEMM2_BF16X=0', 'TACO_DEC_NO_LRES=1',
                                  'TACO_TN_XCD=0', 'TACO_GEMM2_BANK_XCD=0', 'TACO_GEMM2_BSPLIT=0', 'TACO_TN_MERGE_TAPS=0',
                                  'TACO_TAIL_EVENTS=0', 'TACO_TAIL_EVENTS=2', 'TACO_XPROJ_BWD_KSPLIT=0'])
def test_medium_shape_with_optional_paths(built_lib, knob, monkeypatch):
    """The fallback / A-B switches of the train step keep parity: the conv bank's input gradient as K atomic-accumulating problems
    (TACO_NO_BANK_GATHER=1: the path taken when the slabs do not fit or the kernels are not contiguous), gemm2's plain tile order
    (TACO_GEMM2_XCD=0), the fp32 MFMA form of the big GEMMs (TACO_GEMM2_BF16X=0, rounds 2-4), the decoder kernels without
    launch-resident weight rows in LDS (TACO_DEC_NO_LRES=1), the weight-gradient kernels' and the conv banks' plain block orders
    (TACO_TN_XCD=0, TACO_GEMM2_BANK_XCD=0: round 5's XCD-aware orders off) and the weight operand split in registers instead of
    read from the pre-split plane images (TACO_GEMM2_BSPLIT=0, round 6); cross-stream forks / joins through recorded marker packets
    (TACO_TAIL_EVENTS=0, rounds 1-6) or with a stop event on every launch (=2, the learning call's form) instead of the learned
    launch plan; the bi-GRU x-projection's input gradient as one launch (TACO_XPROJ_BWD_KSPLIT=0)."""
    k, v = knob.split('=')
    monkeypatch.setenv(k, v)
    test_medium_shape_forward_backward(built_lib)
    test_backward_without_masks_and_ragged_lengths(built_lib)


def test_tail_event_plan_replays_the_learning_call(built_lib, monkeypatch):
    """Cross-stream forks / joins wait for a stop event that rides on the producing stream's last launch (common.h, layout.hip).
    Which launches carry one is LEARNED: the first call of a kind / shape puts an event on every launch, later calls only on the
    launches a fork / join / gradient segment consumed.  Same inputs, TACO_DETERMINISTIC=1 (fixed summation orders, so a race
    would show as a changed bit): the learning call, three planned calls, a call on recorded markers (TACO_TAIL_EVENTS=0), a call
    that mispredicts (a switch that changes the launch sequence flips between calls: that call falls back to markers, the next
    one learns again) -- outputs, loss and every gradient bit-identical; inference likewise."""
    monkeypatch.setenv('TACO_DETERMINISTIC', '1')
    B, Tt, Td, r, V = 6, 70, 24, 2, 40
    p = on.init_params(V, r, seed=5, perturb=0.3)
    inp, masks = _full_case(B, Tt, Td, r, V, seed_masks=3)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)

    def run():
        R.forward()
        R.backward()
        torch.cuda.synchronize()
        R.check_err()
        return (R.s2s.cpu().numpy().copy(), R.out.cpu().numpy().copy(), R.al.cpu().numpy().copy(), R.loss.cpu().numpy().copy(),
                R.grads.cpu().numpy().copy())

    ref = run()                       # learning call (an event on every launch)
    runs = [run() for _ in range(3)]  # planned calls
    monkeypatch.setenv('TACO_TAIL_EVENTS', '0')
    runs.append(run())                # recorded markers
    monkeypatch.delenv('TACO_TAIL_EVENTS')
    runs.append(run())                # (the plan survives a call that did not use it)
    monkeypatch.setenv('TACO_NO_PRENET_FUSE', '1')   # two launches where the plan expects one: mispredicted forks fall back
    alt = run()
    monkeypatch.delenv('TACO_NO_PRENET_FUSE')
    runs.append(run())                # learns again
    runs.append(run())                # planned again
    for i, got in enumerate(runs):
        for a, b, name in zip(got, ref, ('s2s', 'out', 'align', 'loss', 'grads')):
            assert np.array_equal(a, b), 'run %d: %s differs from the learning call' % (i, name)
    assert np.abs(alt[4].astype(np.float64) - ref[4]).max() <= 1e-4 * np.abs(ref[4]).max()   # (another kernel pair: not bitwise)
    assert np.isfinite(ref[4]).all() and np.abs(ref[4]).max() > 0
    Ri = Runner(built_lib, B, Tt, Td, r, V, train=False)
    Ri.set(p, inp)
    outs = []
    for _ in range(3):
        Ri.infer()
        torch.cuda.synchronize()
        outs.append((Ri.out.cpu().numpy().copy(), Ri.al.cpu().numpy().copy()))
    assert all(np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) for o in outs)
    # a caller that hands in ever new streams (more than the per-thread stream table holds): entries are taken over, results unchanged
    streams = [torch.cuda.Stream() for _ in range(12)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                Ri.infer()
            torch.cuda.synchronize()
            assert np.array_equal(Ri.out.cpu().numpy(), outs[0][0]) and np.array_equal(Ri.al.cpu().numpy(), outs[0][1])


def _full_case(B, Tt, Td, r, V, seed_masks=0):
    from tacotron_amd.data import synthetic_batch
    batch = synthetic_batch(B, Tt, Td, r, V)
    inp = {k: batch[k].numpy() for k in ('text', 'text_length', 'mel', 'stft')}
    rng = np.random.default_rng(seed_masks)
    masks = {'enc_keep1': rng.integers(0, 2, (B, Tt, 256)), 'enc_keep2': rng.integers(0, 2, (B, Tt, 128)),
             'dec_keep1': rng.integers(0, 2, (B, Td, 256)), 'dec_keep2': rng.integers(0, 2, (B, Td, 128)),
             'sample': rng.integers(0, 2, (Td, B))}
    return inp, masks


def _argmax_check(al_hip, al_ref, text_length, min_margin=1e-5):
    """Attention argmax must be bit-exact wherever the oracle's top-1 / top-2 margin is >= min_margin (north_star)."""
    srt = np.sort(al_ref, -1)
    margin = srt[..., -1] - srt[..., -2]
    ok = margin >= min_margin
    same = al_hip.argmax(-1) == al_ref.argmax(-1)
    print('  argmax compared on %d/%d (b,t); min margin overall %.2e, min compared %.2e; max alpha: median %.3f max %.3f; '
          'mismatches among compared: %d, among all: %d' %
          (ok.sum(), ok.size, margin.min(), margin[ok].min() if ok.any() else float('nan'),
           float(np.median(srt[..., -1])), float(srt[..., -1].max()), int((~same & ok).sum()), int((~same).sum())))
    assert np.array_equal(al_hip.argmax(-1)[ok], al_ref.argmax(-1)[ok]), 'attention argmax differs'
    return int(ok.sum())


def l1_tie_adjusted(R, p, inp, masks, r, Td):
    """The loss is a plain L1 sum (tacotron.py:158-160), so its gradient w.r.t. the outputs is sign(output - target): a
    discontinuous function.  Wherever |output - target| is below the forward rounding error (~1e-6) the fp32 HIP path and the
    fp64 oracle may legitimately sit on opposite sides, and ONE such tie changes d loss / d output by 2 in one of 11.8 M
    elements -- 6e-4 of its norm at S1, which the backward pass then carries into every parameter gradient (most visibly into
    those with strong cancellation: embedding, encoder pre_net).  This helper finds those ties with a forward-only oracle run,
    asserts they really are ties (|oracle residual| <= 1e-5), and returns targets nudged by <= 2e-5 at exactly those positions
    so that the oracle's sign pattern equals the HIP path's; gradients can then be compared at the stated tolerance without
    the comparison being decided by a coin flip.  Returns (inputs with adjusted targets, number of ties)."""
    with torch.no_grad():
        pt = ot.to_torch(p, torch.float64)
        ti = {'text': torch.tensor(inp['text'], dtype=torch.int64), 'text_length': torch.tensor(inp['text_length'], dtype=torch.int64),
              'mel': torch.tensor(inp['mel'], dtype=torch.float64), 'stft': torch.tensor(inp['stft'], dtype=torch.float64)}
        if 'speaker' in inp:
            ti['speaker'] = torch.tensor(inp['speaker'], dtype=torch.int64)
        tm = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in (masks or {}).items()}
        s2, o2, _, _ = ot.forward(pt, ti, r, Td, True, tm)
    adj = dict(inp)
    n_ties = 0
    for key, ref, hip in (('mel', s2.numpy(), R.s2s.cpu().numpy()), ('stft', o2.numpy(), R.out.cpu().numpy())):
        tgt = np.asarray(inp[key], dtype=np.float64)
        d_ref, d_hip = ref - tgt, hip.astype(np.float64) - tgt
        ties = np.sign(d_ref) != np.sign(d_hip)
        if ties.any():
            assert np.abs(d_ref[ties]).max() <= 1e-5, 'a sign difference that is not a rounding-level tie: %g' % np.abs(d_ref[ties]).max()
            tgt = tgt.copy()
            tgt[ties] = ref[ties] - np.sign(d_hip[ties]) * 1e-9    # oracle residual takes the HIP path's sign (or 0 where HIP is exactly on target)
            adj[key] = tgt
        n_ties += int(ties.sum())
        print('  L1 sign ties in %s: %d of %d elements (max |oracle residual| at a tie %.1e)' %
              (key, int(ties.sum()), ties.size, float(np.abs(d_ref[ties]).max()) if ties.any() else 0.0))
    return adj, n_ties


def test_full_size_vs_oracle(built_lib):
    """BASELINE configs[1] at FULL size (S1: B=32, Tt=200, Td=180, r=2) against the CPU restatement in fp64: outputs,
    alignments, arg-max on all 5,760 (b,t) whose margin is >= 1e-5, loss, and EVERY parameter gradient (180 steps of BPTT).
    Stated tolerances: outputs rel-L2 <= 1e-4 / max-abs <= 1e-3, alignments max-abs <= 1e-5, loss rel <= 1e-5,
    gradients rel-L2 <= 1e-3 per tensor."""
    B, Tt, Td, r, V = 32, 200, 180, 2, 60
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    inp, masks = _full_case(B, Tt, Td, r, V)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.pb.init_(seed=0)
    p = R.pb.to_dict()
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    adj, n_ties = l1_tie_adjusted(R, p, inp, masks, r, Td)
    lt, s2, o2, a2, ref = ot.loss_and_grads(p, f64(adj), r, Td, f64(masks))
    r1, m1 = report('S1 seq2seq_output', R.s2s.cpu().numpy(), s2)
    r2, m2 = report('S1 output', R.out.cpu().numpy(), o2)
    r3, m3 = report('S1 alignments', R.al.cpu().numpy(), a2)
    assert r1 < 1e-4 and m1 < 1e-3 and r2 < 1e-4 and m2 < 1e-3 and m3 < 1e-5
    loss = R.loss.cpu().numpy()
    print('  loss hip %.3f oracle %.3f' % (loss[0], lt))
    assert abs(loss[0] - lt) <= 1e-5 * lt
    # (at initialisation the softmax over <= 200 positions is nearly flat -- max alpha ~ 0.016 -- so only about a third of
    #  the 5,760 (b,t) clear the 1e-5 margin; the peaked fixture below covers the sharp regime)
    n = _argmax_check(R.al.cpu().numpy(), a2, inp['text_length'])
    assert n > 1000
    # with the L1 ties taken out of the comparison the gradients meet the tolerance of the SMALL cases (2e-4), 5x inside the
    # stated 1e-3 for this size
    bad = check_grads(R, ref, tol=2e-4)
    assert not bad, bad


def test_peaked_attention_fixture(built_lib):
    """Committed medium fixture (B=4, Tt=60, Td=40) whose attention is PEAKED (memory/query/v scaled so max alpha > 0.9 on
    most steps): arg-max bit-exact on every recorded (b,t) with margin >= 1e-5, forward + backward + inference."""
    g = np.load(os.path.join(GOLD, 'model_r2_peaked.npz'))
    r, V, B, Tt, Td = int(g['r']), int(g['V']), int(g['B']), int(g['Tt']), int(g['Td'])
    p = on.init_params(V, r, seed=int(g['seed']), perturb=float(g['perturb']))
    for k, sc in zip(g['scaled_names'], g['scaled_by']):
        p[str(k)] = p[str(k)] * float(sc)
    flat = on.flatten_params(p, V, r, np.float64)
    assert abs(float(np.abs(flat).sum()) - float(g['param_checksum'])) <= 1e-9 * float(g['param_checksum'])
    inp = {'text': g['text'], 'text_length': g['text_length'], 'mel': g['mel'], 'stft': g['stft']}
    masks = {k[5:]: g[k] for k in g.files if k.startswith('mask_')}
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    R.forward()
    al = R.al.cpu().numpy()
    r1, m1 = report('peaked seq2seq_output', R.s2s.cpu().numpy(), g['seq2seq_output'])
    r2, m2 = report('peaked output', R.out.cpu().numpy(), g['output'])
    r3, m3 = report('peaked alignments', al, g['alignments'])
    assert r1 < 1e-5 and m1 < 5e-5 and r2 < 1e-5 and m2 < 5e-5 and m3 < 1e-4   # energies O(1e3): see oracle/make_golden.make_peaked
    assert abs(R.loss[0].item() - float(g['loss'])) <= 1e-5 * float(g['loss'])
    ok = g['argmax_margin'] >= 1e-5
    assert ok.sum() >= 0.95 * ok.size and float((g['alignments'].max(-1) > 0.9).mean()) > 0.5
    assert np.array_equal(al.argmax(-1)[ok], g['argmax'][ok]), 'attention argmax differs'
    print('  peaked: argmax exact on %d/%d (b,t); max alpha > 0.9 on %.0f%% of steps; min compared margin %.2e' %
          (ok.sum(), ok.size, 100 * float((g['alignments'].max(-1) > 0.9).mean()), g['argmax_margin'][ok].min()))
    R.backward()
    _, _, _, _, ref = ot.loss_and_grads(p, f64(inp), r, Td, f64(masks))
    names = [str(n) for n in g['grad_names']]
    for n, gn in zip(names, g['grad_norms']):
        assert abs(np.linalg.norm(ref[n]) - gn) <= 1e-8 * max(1.0, gn)
    bad = check_grads(R, ref)
    assert not bad, bad
    Ri = Runner(built_lib, B, Tt, Td, r, V, train=False)
    Ri.set(p, {'text': inp['text'], 'text_length': inp['text_length']})
    Ri.infer()
    assert report('peaked infer seq2seq_output', Ri.s2s.cpu().numpy(), g['infer_seq2seq_output'])[0] < 1e-5
    ali = Ri.al.cpu().numpy()
    oki = g['infer_argmax_margin'] >= 1e-5
    assert np.array_equal(ali.argmax(-1)[oki], g['infer_argmax'][oki])


def test_s2_envelope(built_lib):
    """BASELINE envelope S2 (B=32, Tt=200, Td=500 = 1000 mel frames, r=2).  Size-independent properties (finite, alignment
    rows sum to 1 and vanish past text_length, loss == fp64 recomputation, bit-reproducible forward, causality: the first
    180 steps equal the S1 run on the truncated inputs) AND a forward comparison with the fp64 CPU restatement."""
    B, Tt, Td, r, V = 32, 200, 500, 2, 60
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    inp, masks = _full_case(B, Tt, Td, r, V)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.pb.init_(seed=0)
    p = R.pb.to_dict()
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    s2s, out, al = R.s2s, R.out, R.al
    assert torch.isfinite(s2s).all() and torch.isfinite(out).all() and torch.isfinite(R.grads).all()
    assert float((al.sum(-1) - 1).abs().max()) < 1e-5
    for b, L in enumerate(inp['text_length']):
        if L < Tt:
            assert float(al[b, :, L:].abs().max()) == 0
    l1 = (s2s.double() - R.mel.double()).abs().sum().item()
    l2 = (out.double() - R.stft.double()).abs().sum().item()
    loss = R.loss.cpu().numpy()
    assert abs(loss[1] - l1) <= 2e-5 * l1 and abs(loss[2] - l2) <= 2e-5 * l2
    s2s0, al0 = s2s.clone(), al.clone()
    R.forward()
    assert torch.equal(s2s0, R.s2s) and torch.equal(al0, R.al)
    # causality: a Td=180 run on the first 180 steps' inputs/masks reproduces the prefix of the 500-step decode
    Rs = Runner(built_lib, B, Tt, 180, r, V)
    ms = {k: (v[:180] if k == 'sample' else (v[:, :180] if k.startswith('dec_') else v)) for k, v in masks.items()}
    Rs.set(p, {'text': inp['text'], 'text_length': inp['text_length'], 'mel': inp['mel'][:, :180], 'stft': inp['stft'][:, :180]}, ms)
    Rs.forward()
    d1 = float((Rs.s2s - s2s0[:, :180]).abs().max())
    d2 = float((Rs.al - al0[:, :180]).abs().max())
    print('  S2 prefix vs S1 run: max|ds2s|=%.2e max|dalign|=%.2e (bitwise: %s)' % (d1, d2, d1 == 0 and d2 == 0))
    # (not bitwise: the pre-net of the teacher-forced steps is a batched GEMM over all B*Td frames, and which GEMM kernel a launch
    #  takes depends on its row count -- 16000 rows here, 5760 in the S1 run -- so p2 differs in the last bits)
    assert d1 <= 5e-6 and d2 <= 1e-6
    del Rs
    # forward vs the CPU restatement (fp64, no autograd)
    with torch.no_grad():
        pt = ot.to_torch(p, torch.float64)
        ti = {'text': torch.tensor(inp['text'], dtype=torch.int64), 'text_length': torch.tensor(inp['text_length'], dtype=torch.int64),
              'mel': torch.tensor(inp['mel'], dtype=torch.float64), 'stft': torch.tensor(inp['stft'], dtype=torch.float64)}
        tm = {k: torch.tensor(v, dtype=torch.float64) for k, v in masks.items()}
        s2, o2, a2, _ = ot.forward(pt, ti, r, Td, True, tm)
    r1, m1 = report('S2 seq2seq_output', s2s0.cpu().numpy(), s2.numpy())
    r2, m2 = report('S2 output', R.out.cpu().numpy(), o2.numpy())
    r3, m3 = report('S2 alignments', al0.cpu().numpy(), a2.numpy())
    assert r1 < 1e-4 and m1 < 1e-3 and r2 < 1e-4 and m2 < 1e-3 and m3 < 1e-5
    _argmax_check(al0.cpu().numpy(), a2.numpy(), inp['text_length'])


@pytest.mark.parametrize('full', [False, True], ids=['medium', 'S1'])
def test_deterministic_gradient_mode(built_lib, full, monkeypatch):
    """TACO_DETERMINISTIC=1: no fp32 atomics decide the order of any gradient sum (split-M weight gradients, K-way conv-bank
    input gradient, BN / bias column sums, embedding scatter, attention_v) -- two runs of the same step agree bit for bit, and
    the gradients still match the default mode to rounding."""
    if full:
        B, Tt, Td, r, V = 32, 200, 180, 2, 60
        inp, masks = _full_case(B, Tt, Td, r, V)
    else:
        r, V, B, Tt, Td = 2, 40, 4, 37, 12
        inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=8)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.pb.init_(seed=3)
    p = R.pb.to_dict()
    R.set(p, inp, masks)
    R.forward()
    R.backward()
    g_default = R.grads.clone()
    monkeypatch.setenv('TACO_DETERMINISTIC', '1')
    runs = []
    for _ in range(2):
        R.forward()
        R.backward()
        runs.append((R.grads.clone(), R.loss.clone(), R.s2s.clone()))
    assert torch.equal(runs[0][0], runs[1][0]), 'gradients differ between two deterministic runs'
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    d = float((runs[0][0] - g_default).norm() / g_default.norm())
    print('  deterministic vs default gradients: rel-L2 %.2e' % d)
    assert d < 1e-5


def test_model_class_train_steps_reduce_loss(built_lib):
    """Host mirror of the reference object: Tacotron(config, inputs, train).step(lr) runs and learns."""
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.model import Tacotron
    c = Config()
    c.r, c.vocab_size = 2, 30
    batch = synthetic_batch(4, 24, 10, 2, 30, seed=5, min_len=10)
    m = Tacotron(c, batch, train=True, seed=1)
    losses = []
    for _ in range(8):
        m.step(lr=1e-3)
        losses.append(float(m.loss))
    m.check()   # no decoder exchange time-out was flagged
    print('  losses', ['%.1f' % l for l in losses], 'gnorm', float(m.global_gradient_norm))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert m.global_step == 8
    # inference object over the same parameters (test.py contract)
    c.max_decode_iter = 10
    mi = Tacotron(c, {'text': batch['text'], 'text_length': batch['text_length']}, train=False, params=m.params)
    out, al = mi.run()
    assert out.shape == (4, 10, 2050) and al.shape == (4, 10, 24) and torch.isfinite(out).all()
    # checkpoint round trip
    sd = m.state_dict()
    m2 = Tacotron(c, batch, train=True, seed=2)
    m2.load_state_dict(sd)
    assert torch.equal(m2.params.flat, m.params.flat) and m2.global_step == 8


def test_twenty_step_trajectory_vs_oracle(built_lib):
    """tacotron.py:167-185 end to end through the host object (VERDICT r4 #8a): 20 x Tacotron.step() -- masks drawn on the device
    by taco_fill_bernoulli and READ BACK, forward, backward, global-norm clip, TF-form Adam with its bias correction, global_step.
    Every step is checked on both sides of the optimizer:
      * loss and every parameter gradient against the fp64 oracle evaluated AT THE DEVICE'S CURRENT PARAMETERS with the masks the
        device drew (pins which byte buffer feeds which layer, at every step of a moving trajectory);
      * the parameters after the update against the oracle's clip + Adam (fp64) applied to the DEVICE'S gradients, with the
        oracle's m / v slots carried from step to step (pins the step counter of the bias correction, the clip threshold, the
        update form and that global_step / the slots persist across steps).
    A free-running fp64 trajectory is NOT the reference: Adam's m / (sqrt(v) + eps) gives every element whose gradient is below the
    fp32 noise floor an lr-sized update of rounding-determined sign, so an fp32 and an fp64 run drift apart (first version of this
    test: global norm 2.5e-4 apart at step 6, 1.2e-3 at step 9) without either being wrong.
    Tolerances: loss rel 1e-5; the WHOLE gradient vector rel-L2 1e-3 (the suite's 2e-4 per tensor holds while every ReLU / max-pool
    decision agrees with fp64; over 20 steps of a moving trajectory at this tiny shape a near-tie flips now and then -- step 13 of
    this seed: 3.1e-4 on the whole vector, 1.4e-5 at every other step: DESIGN.md 1 (vi); a misrouted mask moves it by tens of percent); global norm rel 1e-5; parameters 2e-7 + 1e-3 lr abs."""
    from oracle import taco_torch as ot
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.model import Tacotron
    c = Config()
    c.r, c.vocab_size = 2, 20
    B, Tt, Td = 3, 14, 7
    batch = synthetic_batch(B, Tt, Td, c.r, c.vocab_size, seed=9, min_len=6)
    m = Tacotron(c, batch, train=True, seed=4)
    table = [(n, off, size, dims) for n, off, size, dims in m.params.table]
    p0 = {k: v.astype(np.float64) for k, v in m.params.to_dict().items()}
    mt = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in p0.items()}
    vt = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in p0.items()}
    inp = {k: batch[k].numpy() for k in ('text', 'text_length', 'mel', 'stft')}
    lr = 1e-3
    worst = {'loss': 0.0, 'grad': 0.0, 'gnorm': 0.0, 'param': 0.0}
    for step in range(1, 21):
        before = {k: v.astype(np.float64) for k, v in m.params.to_dict().items()}
        masks = m.draw_masks()
        m.forward(masks)
        m.backward()
        gflat = m.grads.detach().cpu().numpy().astype(np.float64)
        g_hip = {n: gflat[off:off + size].reshape(dims) for n, off, size, dims in table}
        m.apply_gradients(lr)
        assert m.global_step == step
        # (1) the device's loss and gradients at ITS parameters, with ITS masks
        fm = {k: v.cpu().numpy().astype(np.float64) for k, v in masks.items()}
        loss, _, _, _, grads = ot.loss_and_grads(before, inp, c.r, Td, fm)
        worst['loss'] = max(worst['loss'], abs(float(m.loss) - loss) / loss)
        assert abs(float(m.loss) - loss) <= 1e-5 * loss, (step, float(m.loss), loss)
        num = den = 0.0
        for k, g in grads.items():
            g = np.zeros_like(before[k]) if g is None else g
            num += float(((g_hip[k] - g) ** 2).sum())
            den += float((g ** 2).sum())
        d = (num / den) ** 0.5
        worst['grad'] = max(worst['grad'], d)
        assert d <= 1e-3, (step, d)
        # (2) the device's update against the oracle optimizer fed the device's gradients
        pt = {k: torch.tensor(v) for k, v in before.items()}
        gn = ot.clip_adam_step(pt, {k: torch.tensor(v) for k, v in g_hip.items()}, mt, vt, step, lr)
        worst['gnorm'] = max(worst['gnorm'], abs(float(m.global_gradient_norm) - gn) / gn)
        assert abs(float(m.global_gradient_norm) - gn) <= 1e-5 * gn, (step, float(m.global_gradient_norm), gn)
        after = m.params.to_dict()
        for k, v in pt.items():
            d = np.abs(after[k].astype(np.float64) - v.numpy()).max()
            worst['param'] = max(worst['param'], d)
            assert d <= 2e-7 + 1e-3 * lr, (step, k, d)
    m.check()
    moved = max(np.abs(m.params.to_dict()[k].astype(np.float64) - p0[k]).max() for k in p0)
    print('  20 steps: worst loss rel %.2e, gradient rel-L2 %.2e, global norm rel %.2e, parameter abs %.2e; largest movement %.2e' %
          (worst['loss'], worst['grad'], worst['gnorm'], worst['param'], moved))
    assert moved > 5 * lr   # (20 lr-sized Adam updates, not a no-op)


def test_error_words_are_sticky_and_guard_the_update(built_lib):
    """A decoder exchange time-out must not reach the parameters: with an error word set, clip+Adam skips itself on the
    device (gnorm = -1), the flag survives further forward passes until check() raises and clears it."""
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.model import Tacotron
    c = Config()
    c.r, c.vocab_size = 2, 30
    batch = synthetic_batch(4, 24, 10, 2, 30, seed=5, min_len=10)
    m = Tacotron(c, batch, train=True, seed=1)
    m.step(lr=1e-3)
    torch.cuda.synchronize()
    m.check()
    assert float(m.global_gradient_norm) > 0
    before = m.params.flat.clone()
    m_before = m.adam_m.clone()
    m._err[1] = 1                      # what decoder_bwd_kernel does on a timed-out exchange
    m.step(lr=1e-3)
    m.forward(m.draw_masks())          # a later forward must NOT erase the flag
    torch.cuda.synchronize()
    assert torch.equal(m.params.flat, before) and torch.equal(m.adam_m, m_before)
    assert float(m.global_gradient_norm) == -1.0
    assert built_lib.decoder_mode() == 0
    try:
        with pytest.raises(built_lib.TacoError) as ei:
            m.check()
        # a FIRST time-out is recoverable and does not degrade the process (one transient stall must not cost the rest of
        # a multi-day run its fast decoder mode) ...
        assert ei.value.recoverable and built_lib.decoder_mode() == 0
        m.check()                          # cleared by the raise above
        m.step(lr=1e-3)
        torch.cuda.synchronize()
        assert float(m.global_gradient_norm) > 0 and m.placement_census()[1] == 0
        # ... a second one within Tacotron.ESCALATE_WINDOW steps moves the process to the next more conservative mode
        before = m.params.flat.clone()
        m._err[1] = 1
        with pytest.raises(built_lib.TacoError) as ei:
            m.check()
        assert ei.value.recoverable and built_lib.decoder_mode() == 1 and m.decoder_mode == 1
        m.check()
        m.step(lr=1e-3)                    # ... and the next step runs in that mode (agent-scope exchange) and updates again
        torch.cuda.synchronize()
        assert not torch.equal(m.params.flat, before) and float(m.global_gradient_norm) > 0
        assert m.placement_census()[0] == 0 and m.placement_census()[1] > 0
        m.check()
        # a second and a third time-out: decoder.hip, then nothing left to fall back to
        m._err[0] = 1
        with pytest.raises(built_lib.TacoError) as ei:
            m.check()
        assert ei.value.recoverable and built_lib.decoder_mode() == 2
        m.step(lr=1e-3)
        torch.cuda.synchronize()
        assert float(m.global_gradient_norm) > 0 and built_lib.last_cluster(0) < 32
        m._err[0] = 1
        with pytest.raises(built_lib.TacoError) as ei:
            m.check()
        assert not ei.value.recoverable and built_lib.decoder_mode() == 2
    finally:
        built_lib.decoder_mode(0)
    # error word 2 = "cluster not co-resident" (the placement rendezvous of decoder3.hip): reported as such, and the process goes
    # straight to decoder.hip -- both decoder3 modes need the residency that was just found missing
    try:
        m.check()
        m._last_timeout_step = None
        m._err[0] = 2
        with pytest.raises(built_lib.TacoError) as ei:
            m.check()
        assert ei.value.recoverable and ei.value.not_resident and 'NOT CO-RESIDENT' in str(ei.value)
        assert built_lib.decoder_mode() == 2
        m.step(lr=1e-3)
        torch.cuda.synchronize()
        m.check()
        assert float(m.global_gradient_norm) > 0 and built_lib.last_cluster(0) < 32
    finally:
        built_lib.decoder_mode(0)
    # a batch of another shape is refused instead of being read out of bounds
    with pytest.raises(built_lib.TacoError):
        m.set_inputs(synthetic_batch(4, 25, 10, 2, 30, seed=5, min_len=10))


def test_inference_is_graph_capturable(built_lib):
    """One `taco_infer` call is a pure stream-ordered enqueue (no allocation, no host synchronisation, side-stream fork/join
    by events only): it can be captured into a HIP graph and the replay reproduces the eager result bit for bit."""
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.model import Tacotron
    c = Config()
    c.r, c.vocab_size, c.max_decode_iter = 2, 30, 6
    b = synthetic_batch(2, 24, 6, 2, 30, seed=3, min_len=8)
    m = Tacotron(c, b, train=False, seed=5)
    m.run()
    torch.cuda.synchronize()
    ref_out, ref_al = m.output.clone(), m.alignments.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m.run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            m.run()
    torch.cuda.synchronize()
    m.output.zero_()
    m.alignments.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(m.output, ref_out) and torch.equal(m.alignments, ref_al)
    m.check()


@pytest.mark.parametrize('B,mode,r', [(11, 'default', 2), (12, 'agent', 2), (20, 'default', 2), (5, 'v3_off', 2), (32, 'agent', 2),
                                      (20, 'default', 5), (9, 'default', 5), (40, 'default', 2), (48, 'default', 2), (50, 'default', 2), (64, 'default', 2),
                                      (70, 'default', 5)])
def test_decoder3_cluster_geometries(built_lib, B, mode, r, monkeypatch):
    """decoder3.hip (clusters of 32 workgroups x R rows, register-resident weights): R = 1 / 2 / 4 rows per cluster incl. a
    partially filled last cluster (B = 11: six clusters of two rows, the last with one valid row; B = 20: five clusters of
    four), the placement-independent agent-scope exchange forced (TACO_DEC_V3_AGENT=1: what a cluster that straddles XCDs
    uses), and the decoder.hip fall-back (TACO_DEC_V3=0) -- forward, backward and inference against the fp64 restatement.
    r = 5 (BASELINE configs[0]'s reduction factor) at four and two rows per cluster: the widest instantiations of both kernels
    (400 output columns per step; 7 d-out prefetch jobs per loader thread, 252 VGPRs in the backward kernel).  B > 32 (round 6): the
    batch runs as consecutive decoder3 launches of <= 32 rows (B = 40: 32 rows at R = 4 + 8 at R = 1; B = 50: 32 + 18 at R = 4
    with a half-filled last cluster; B = 70: 32 + 32 + 6) instead of falling to decoder.hip."""
    if mode == 'agent':
        monkeypatch.setenv('TACO_DEC_V3_AGENT', '1')
    if mode == 'v3_off':
        monkeypatch.setenv('TACO_DEC_V3', '0')
    V, Tt, Td = 33, 41, 9
    p = on.init_params(V, r, seed=8, perturb=0.2)
    inp, masks = small_case(r=r, V=V, B=B, Tt=Tt, Td=Td, seed=40 + B)
    R = Runner(built_lib, B, Tt, Td, r, V)
    R.set(p, inp, masks)
    R.forward()
    assert built_lib.last_cluster(0) == (32 if mode != 'v3_off' else 8)
    R.backward()
    assert built_lib.last_cluster(1) == (32 if mode != 'v3_off' else 8)
    adj, _ = l1_tie_adjusted(R, p, inp, masks, r, Td)
    lt, s2, o2, a2, ref = ot.loss_and_grads(p, f64(adj), r, Td, f64(masks))
    assert report('s2s (B=%d, %s)' % (B, mode), R.s2s.cpu().numpy(), s2)[0] < 1e-5
    assert report('out', R.out.cpu().numpy(), o2)[0] < 1e-5
    assert report('align', R.al.cpu().numpy(), a2)[1] < 1e-6
    assert abs(R.loss[0].item() - lt) <= 1e-5 * lt
    bad = check_grads(R, ref)
    if bad and B > 32:
        # more rows = more ReLU / max-pool decisions = a fair chance that fp32 and fp64 take ONE of them differently (a pre-activation
        # within rounding of its boundary), which moves the small tensors below it by a few 1e-4 (tests/test_gpu_sizes.py exhibits
        # this at full size).  The flips are read back from the workspace, each must sit within rounding of its boundary, every
        # DECODER tensor must meet the tolerance as it is, and with the HIP path's decisions imposed on the fp64 graph every tensor does.
        from tests.decisions import as_force, flips, hip_decisions
        hip, ok, how = hip_decisions(R, p, masks, B, Tt, Td, r, 1)
        dec = ot.Decisions()
        ot.loss_and_grads(p, f64(adj), r, Td, f64(masks), dec=dec)
        fl = flips(hip, ok, dec)
        print('  B=%d: %d decision flip(s) vs fp64: %s; tensors off without forcing: %s' % (B, len(fl), fl[:6], bad))
        assert 1 <= len(fl) <= 16 and all(mg <= 1e-5 for _, _, mg in fl), fl
        assert not [n for n, _ in bad if n.startswith('decoder')], bad
        ref_forced = ot.loss_and_grads(p, f64(adj), r, Td, f64(masks), dec=ot.Decisions(as_force(hip)))[4]
        bad = check_grads(R, ref_forced)
    assert not bad, bad
    Ri = Runner(built_lib, B, Tt, Td, r, V, train=False)
    Ri.set(p, {'text': inp['text'], 'text_length': inp['text_length']})
    Ri.infer()
    si, oi, ai = on.forward(p, f64(inp), r, Td, train=False, masks=None)[:3]
    assert report('infer s2s', Ri.s2s.cpu().numpy(), si)[0] < 1e-5
    assert report('infer align', Ri.al.cpu().numpy(), ai)[1] < 1e-6
