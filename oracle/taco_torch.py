"""Second, independently written CPU restatement of the reference model, on torch-CPU primitives with
autograd -- TEST INFRASTRUCTURE ONLY (gradient oracle + `cpu_baseline` timing leg of bench.py).

PARITY UNPINNED (see oracle/taco_numpy.py header): TensorFlow 1.2 cannot run here.  This file exists so
that two implementations written against different primitives (NumPy shifted-matmul convs / manual
softmax there; F.conv1d / F.max_pool1d / torch.softmax / torch.bmm + autograd here) must agree before
either is used to judge the HIP path.

Follows /root/reference/models/tacotron.py:35-195 and /root/reference/models/ops.py:5-132; parameter
names/layouts are those of oracle/taco_numpy.py::param_spec (TF variable layouts).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3


class Decisions:
    """The model's DISCRETE decisions -- which side of a ReLU a pre-activation falls on, which of the two rows a max-pool
    keeps -- recorded (with their margins) or forced.  fp32 and fp64 evaluate the same graph to ~1e-7, but a pre-activation
    that is 1e-8 away from zero can land on different sides, and the gradient below that unit then differs by O(1) in its
    receptive field.  tests/test_gpu_sizes.py uses this to EXHIBIT such flips (HIP decision != fp64 decision, margin at
    rounding level) and to compare gradients with the HIP path's decisions imposed on the fp64 graph.
      record mode (force=None): `rec[name]` bool tensor, `margin[name]` |distance to the decision boundary|
      force mode: `force[name]` float 0/1 tensor replaces the decision at that site (sites not in `force` are recorded)."""

    def __init__(self, force=None):
        self.force = force or {}
        self.rec, self.margin = {}, {}

    def relu(self, name, x):
        if name in self.force:
            return x * self.force[name]
        self.rec[name] = (x > 0).detach()
        self.margin[name] = x.abs().detach()
        return torch.relu(x)

    def pool2(self, name, z):
        """max(z[t], z[t+1]) along dim 1, the last row alone; decision True = row t+1 wins (strictly: a tie keeps row t, as
        torch.max_pool1d's and the HIP kernels' backward do)."""
        nxt = torch.cat([z[:, 1:], z[:, -1:]], 1)
        if name in self.force:
            m = self.force[name]
            return torch.where(m > 0.5, nxt, z)
        take = nxt > z
        take[:, -1] = False
        self.rec[name] = take.detach()
        mg = (nxt - z).abs().detach()
        mg[:, -1] = float('inf')
        self.margin[name] = mg
        return torch.where(take, nxt, z)


def _relu(dec, name, x):
    return torch.relu(x) if dec is None else dec.relu(name, x)


def _conv_same(x, kernel, bias):
    # x (B,T,Cin); kernel TF layout (k,Cin,Cout) -> torch (Cout,Cin,k).  ops.py:54-60
    k = kernel.shape[0]
    pl = (k - 1) // 2
    pr = k - 1 - pl
    xt = F.pad(x.transpose(1, 2), (pl, pr))
    y = F.conv1d(xt, kernel.permute(2, 1, 0), bias)
    return y.transpose(1, 2)


def _bn(x, gamma, beta):
    # ops.py:64,87 -- inference-mode BN with moving mean 0 / var 1
    return x * (gamma * (1.0 / math.sqrt(1.0 + BN_EPS))) + beta


def _maxpool(x):
    # ops.py:66-71 -- pool 2, stride 1, SAME (pad after with -inf)
    xt = F.pad(x.transpose(1, 2), (0, 1), value=float('-inf'))
    return F.max_pool1d(xt, 2, 1).transpose(1, 2)


def _gru(x, h, wg, bg, wc, bc):
    H = h.shape[-1]
    g = torch.sigmoid(torch.addmm(bg, torch.cat([x, h], 1), wg))
    r, u = g.split(H, 1)
    c = torch.tanh(torch.addmm(bc, torch.cat([x, r * h], 1), wc))
    return u * h + (1 - u) * c


def _bigru(x, p, prefix, h0=None):
    B, T, _ = x.shape
    H = p[prefix + 'fw/gates/bias'].shape[0] // 2
    outs = []
    for name, order in (('fw', range(T)), ('bw', range(T - 1, -1, -1))):
        wg, bg = p[prefix + name + '/gates/kernel'], p[prefix + name + '/gates/bias']
        wc, bc = p[prefix + name + '/candidate/kernel'], p[prefix + name + '/candidate/bias']
        h = x.new_zeros(B, H) if h0 is None else h0
        seq = [None] * T
        for t in order:
            h = _gru(x[:, t], h, wg, bg, wc, bc)
            seq[t] = h
        outs.append(torch.stack(seq, 1))
    return torch.cat(outs, 2)


def _highway(x, p, prefix, dec=None):
    if (prefix + 'adapt/kernel') in p:
        x = F.linear(x, p[prefix + 'adapt/kernel'].t(), p[prefix + 'adapt/bias'])
    t = torch.sigmoid(F.linear(x, p[prefix + 'T/kernel'].t(), p[prefix + 'T/bias']))
    h = _relu(dec, prefix + 'H', F.linear(x, p[prefix + 'H/kernel'].t(), p[prefix + 'H/bias']))
    return h * t + x * (1 - t)


def cbhg(x, p, prefix, K, spk=None, dec=None):
    bank = torch.cat([_conv_same(x, p[prefix + 'bank_%d/kernel' % k], p[prefix + 'bank_%d/bias' % k])
                      for k in range(1, K + 1)], 2)
    bank = _relu(dec, prefix + 'bank', bank)
    z = _bn(bank, p[prefix + 'bank_bn/gamma'], p[prefix + 'bank_bn/beta'])
    y = _maxpool(z) if dec is None else dec.pool2(prefix + 'pool', z)
    y = _relu(dec, prefix + 'proj1', _conv_same(y, p[prefix + 'proj1/kernel'], p[prefix + 'proj1/bias']))
    y = _bn(y, p[prefix + 'proj1_bn/gamma'], p[prefix + 'proj1_bn/beta'])
    y = _conv_same(y, p[prefix + 'proj2/kernel'], p[prefix + 'proj2/bias'])
    y = _bn(y, p[prefix + 'proj2_bn/gamma'], p[prefix + 'proj2_bn/beta'])
    h = y + x
    for l in range(4):
        hp = prefix + 'highway_%d/' % l
        if spk is not None:   # ops.py:101-105
            sv = _relu(dec, hp + 'spk', F.linear(spk, p[hp + 'spk/kernel'].t(), p[hp + 'spk/bias']))
            h = torch.cat([h, sv[:, None, :].expand(-1, h.shape[1], -1)], 2)
        h = _highway(h, p, hp, dec)
    h0 = None
    if spk is not None:       # ops.py:111-115
        h0 = _relu(dec, prefix + 'gru_init', F.linear(spk, p[prefix + 'gru_init/kernel'].t(), p[prefix + 'gru_init/bias']))
    return _bigru(h, p, prefix + 'bigru/', h0)


def _prenet(x, p, prefix, k1, k2, dec=None, tag=''):
    l1 = _relu(dec, prefix + 'l1' + tag, F.linear(x, p[prefix + 'dense/kernel'].t(), p[prefix + 'dense/bias']))
    if k1 is not None:
        l1 = l1 * (2.0 * k1)
    l2 = _relu(dec, prefix + 'l2' + tag, F.linear(l1, p[prefix + 'dense_1/kernel'].t(), p[prefix + 'dense_1/bias']))
    if k2 is not None:
        l2 = l2 * (2.0 * k2)
    return l2


def forward(p, inputs, r, n_steps, train, masks=None, dec=None):
    """p: dict name->tensor; inputs: dict of tensors (text int64, text_length int64, mel, stft);
    masks: dict of float tensors (0/1); dec: optional Decisions (record / force the ReLU and max-pool decisions; the decoder
    pre_net sites are named per step: 'decoder/pre_net/l1@<t>').  Returns (seq2seq_output, output, alignments, encoded)."""
    masks = masks or {}
    g = (lambda k: masks.get(k)) if train else (lambda k: None)
    text = inputs['text']
    B, Tt = text.shape
    emb = F.embedding(text, p['embedding'])
    spk = F.embedding(inputs['speaker'], p['speaker_embed']) if ('speaker' in inputs and 'speaker_embed' in p) else None
    enc = cbhg(_prenet(emb, p, 'encoder/pre_net/', g('enc_keep1'), g('enc_keep2'), dec), p, 'encoder/cbhg/', 16, spk, dec)

    # attention memory (tacotron.py:48-52)
    valid = torch.arange(Tt)[None, :] < inputs['text_length'][:, None]
    values = enc * valid[:, :, None].to(enc.dtype)
    keys = torch.matmul(values, p['decoder/memory_layer/kernel'])
    v = p['decoder/attention_v']

    nmel = 80
    h = [enc.new_zeros(B, 256) for _ in range(3)]
    att = enc.new_zeros(B, 256)
    mel = inputs.get('mel') if train else None
    prev = mel[:, 0] if mel is not None else enc.new_zeros(B, nmel * r)
    outs, aligns = [], []
    dk1, dk2, smp = g('dec_keep1'), g('dec_keep2'), g('sample')
    for t in range(n_steps):
        pn = _prenet(prev[:, nmel * (r - 1):], p, 'decoder/pre_net/',
                     dk1[:, t] if dk1 is not None else None, dk2[:, t] if dk2 is not None else None, dec, '@%d' % t)
        x = F.linear(torch.cat([pn, att], 1), p['decoder/in_proj/kernel'].t(), p['decoder/in_proj/bias'])
        inp = x
        for l in range(3):
            h[l] = _gru(inp, h[l], p['decoder/gru_%d/gates/kernel' % l], p['decoder/gru_%d/gates/bias' % l],
                        p['decoder/gru_%d/candidate/kernel' % l], p['decoder/gru_%d/candidate/bias' % l])
            inp = h[l]
        o = F.linear(x + h[2], p['decoder/out_proj/kernel'].t(), p['decoder/out_proj/bias'])
        q = torch.mm(o, p['decoder/query_layer/kernel'])
        score = torch.tanh(keys + q[:, None, :]).matmul(v)
        score = score.masked_fill(~valid, float('-inf'))
        a = torch.softmax(score, 1)
        ctx = torch.bmm(a[:, None, :], values)[:, 0]
        att = torch.mm(torch.cat([o, ctx], 1), p['decoder/attention_layer/kernel'])
        outs.append(o)
        aligns.append(a)
        if mel is None:
            prev = o
        elif t + 1 < n_steps:
            prev = mel[:, t + 1]
            if smp is not None:
                prev = torch.where(smp[t][:, None] > 0.5, o, prev)
    s2s = torch.stack(outs, 1)
    al = torch.stack(aligns, 1)
    post = cbhg(s2s.reshape(B, n_steps * r, nmel), p, 'post/cbhg/', 8, None, dec)
    out = F.linear(post, p['post/dense/kernel'].t(), p['post/dense/bias']).reshape(B, n_steps, -1)
    return s2s, out, al, enc


def loss_fn(s2s, out, mel, stft):
    return (s2s - mel).abs().sum() + (out - stft).abs().sum()


def to_torch(pnp, dtype=torch.float64, requires_grad=False):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in pnp.items()}


def loss_and_grads(pnp, inputs_np, r, n_steps, masks_np=None, dtype=torch.float64, dec=None):
    """Convenience: numpy in, numpy out.  Returns (loss, s2s, out, align, grads dict).  dec: optional Decisions."""
    p = to_torch(pnp, dtype, True)
    inputs = {
        'text': torch.tensor(inputs_np['text'], dtype=torch.int64),
        'text_length': torch.tensor(inputs_np['text_length'], dtype=torch.int64),
        'mel': torch.tensor(inputs_np['mel'], dtype=dtype),
        'stft': torch.tensor(inputs_np['stft'], dtype=dtype),
    }
    if 'speaker' in inputs_np:
        inputs['speaker'] = torch.tensor(inputs_np['speaker'], dtype=torch.int64)
    masks = {k: torch.tensor(v, dtype=dtype) for k, v in (masks_np or {}).items()}
    s2s, out, al, _ = forward(p, inputs, r, n_steps, True, masks, dec)
    loss = loss_fn(s2s, out, inputs['mel'], inputs['stft'])
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else None) for k, t in p.items()}
    return float(loss.detach()), s2s.detach().numpy(), out.detach().numpy(), al.detach().numpy(), grads


def clip_adam_step(params, grads, m, v, step, lr, cap=5.0, b1=0.9, b2=0.999, eps=1e-8):
    """tacotron.py:167-185 on torch tensors (in place).  Returns global norm."""
    with torch.no_grad():
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
        scale = cap / max(gn, cap) if cap > 0 else 1.0
        lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        for k in params:
            g = grads[k] * scale
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            params[k].sub_(lr_t * m[k] / (v[k].sqrt() + eps))
    return gn
