"""CPU restatement (NumPy) of the reference Tacotron acoustic model -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference is a TensorFlow-1.2 graph (`/root/reference/models/tacotron.py`,
`/root/reference/models/ops.py`).  TensorFlow 1.2 cannot be installed or imported in this image and the
reference ships no golden vectors / tests for this path, so this restatement is
written from the reference's call sites plus the published TF-r1.2 semantics of the contrib ops they
invoke.  It is cross-checked against a second, independently written implementation
(`oracle/taco_torch.py`) and each TF-semantics assumption is listed in ASSUMPTIONS below so that a
future TF-1.x run can falsify it.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product path (`tacotron_amd/`) never does.

All tensors are batch-major row-major `(B, T, C)`; weights use TF variable layouts
(dense kernel `(in, out)`, conv1d kernel `(k, Cin, Cout)`, GRU gates kernel `(Cin+H, 2H)` ...).
Works in whatever float dtype the parameters are given in (float64 for goldens, float32 for timing).
"""
from __future__ import annotations

import numpy as np

ASSUMPTIONS = [
    "A1 tf.layers.batch_normalization called with training=False and never-updated moving stats "
    "(mean 0, var 1): y = gamma * x / sqrt(1 + 1e-3) + beta           (ops.py:64,87)",
    "A2 conv1d padding='same', stride 1: pad_left=(k-1)//2, pad_right=k-1-pad_left, "
    "cross-correlation, kernel layout (k, Cin, Cout)                   (ops.py:54-60,80-86)",
    "A3 max_pooling1d(pool=2, stride=1, 'same'): y[t]=max(x[t],x[t+1]), y[T-1]=x[T-1] (ops.py:66-71)",
    "A4 GRUCell r1.2: [r,u]=sigmoid([x,h]Wg+bg) (bias init 1.0, split r then u); "
    "c=tanh([x,r*h]Wc+bc); h'=u*h+(1-u)*c                            (ops.py:118-119, tacotron.py:54)",
    "A5 decoder cell = OutputProjection(InputProjection(Residual(MultiRNN[GRU x3]),256),80r), both "
    "projections with bias, ONE residual around the 3-GRU stack        (tacotron.py:54-60)",
    "A6 BahdanauAttention r1.2: values=memory masked to 0 past text_length, keys=values Wm (no bias), "
    "score=sum_u v_u tanh(keys+query Wq), -inf past text_length, softmax (tacotron.py:48-52)",
    "A7 AttentionWrapper r1.2: query = cell_output (the projected 80r mel frame group), "
    "attention = [cell_output; context] Wa (no bias), output_attention=False (tacotron.py:73-80)",
    "A8 Training helper feeds mel[:,t] at step t (unshifted), sampled rows (Bernoulli per row per step) "
    "feed cell_output[t] into step t+1, gradient flows through them    (tacotron.py:82-87)",
    "A9 InferenceHelper: zeros first input, next input = cell_output, always max_decode_iter steps "
    "                                                                  (ops.py:5-25)",
    "A10 dropout(rate=0.5, training): keep mask * 2 (inverted dropout) (tacotron.py:41-43)",
    "A11 loss = sum|seq2seq_output-mel| + sum|output-stft| (no mask)    (tacotron.py:158-160)",
    "A13 multi-speaker (num_speakers>1): speaker table glorot-init; encoder CBHG only: per highway layer "
    "s=relu(dense(spk,128)) concatenated to h (highway adapts 256->128), GRU h0=relu(dense(spk,128)) for both "
    "directions; post-net CBHG and decoder get no speaker input      (ops.py:101-127, tacotron.py:96-97,131,147)",
    "A12 Adam TF form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps); "
    "clip_by_global_norm(5): g *= 5/max(||g||,5)                      (tacotron.py:170-184)",
]

BN_EPS = 1e-3


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def relu(x):
    return np.maximum(x, 0)


def dense(x, kernel, bias=None):
    """tf.layers.dense: x (..., in) @ kernel (in, out) + bias."""
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return y


def conv1d_same(x, kernel, bias):
    """tf.layers.conv1d(padding='same'), stride 1 (ops.py:54-60, 80-86).  x (B,T,Cin), kernel (k,Cin,Cout)."""
    k = kernel.shape[0]
    B, T, _ = x.shape
    pad_l = (k - 1) // 2
    pad_r = k - 1 - pad_l
    xp = np.pad(x, ((0, 0), (pad_l, pad_r), (0, 0)))
    y = np.zeros((B, T, kernel.shape[2]), dtype=x.dtype)
    for j in range(k):
        y += xp[:, j:j + T, :] @ kernel[j]
    return y + bias


def bn_affine(x, gamma, beta):
    """tf.layers.batch_normalization in inference mode with mean 0 / var 1 (ops.py:64,87)."""
    scale = gamma / np.sqrt(np.asarray(1.0 + BN_EPS, dtype=x.dtype))
    return x * scale + beta


def maxpool2_same(x):
    """tf.layers.max_pooling1d(pool_size=2, strides=1, padding='same') (ops.py:66-71)."""
    y = x.copy()
    y[:, :-1, :] = np.maximum(x[:, :-1, :], x[:, 1:, :])
    return y


def gru_cell(x, h, wg, bg, wc, bc):
    """tf.contrib.rnn.GRUCell r1.2.  wg (Cin+H, 2H), wc (Cin+H, H).  Returns (h', r, u, c)."""
    H = h.shape[-1]
    g = sigmoid(np.concatenate([x, h], -1) @ wg + bg)
    r, u = g[..., :H], g[..., H:]
    c = np.tanh(np.concatenate([x, r * h], -1) @ wc + bc)
    return u * h + (1 - u) * c, r, u, c


def bigru(x, p, prefix, h0=None):
    """tf.nn.bidirectional_dynamic_rnn(GRUCell, GRUCell, x) without sequence_length (ops.py:117-128).
    h0 (B,H): initial state of BOTH directions (ops.py:123-124, the speaker site), default zeros."""
    B, T, _ = x.shape
    H = p[prefix + 'fw/gates/bias'].shape[0] // 2
    out = np.zeros((B, T, 2 * H), dtype=x.dtype)
    for d, name in enumerate(('fw', 'bw')):
        wg, bg = p[prefix + name + '/gates/kernel'], p[prefix + name + '/gates/bias']
        wc, bc = p[prefix + name + '/candidate/kernel'], p[prefix + name + '/candidate/bias']
        h = np.zeros((B, H), dtype=x.dtype) if h0 is None else h0.copy()
        ts = range(T) if d == 0 else range(T - 1, -1, -1)
        for t in ts:
            h, _, _, _ = gru_cell(x[:, t, :], h, wg, bg, wc, bc)
            out[:, t, d * H:(d + 1) * H] = h
    return out


def highway(x, p, prefix):
    """ops.highway (ops.py:27-46): optional adapting dense, then T/H gates."""
    if (prefix + 'adapt/kernel') in p:
        x = dense(x, p[prefix + 'adapt/kernel'], p[prefix + 'adapt/bias'])
    t = sigmoid(dense(x, p[prefix + 'T/kernel'], p[prefix + 'T/bias']))
    h = relu(dense(x, p[prefix + 'H/kernel'], p[prefix + 'H/bias']))
    return h * t + x * (1 - t)


def cbhg(x, p, prefix, K, speaker_embed=None):
    """ops.CBHG (ops.py:48-132).  speaker_embed (B,16) enables the Deep-Voice-2 style sites (ops.py:101-115): per highway
    layer s = relu(dense(spk, C)) tiled over T and concatenated to h (so highway() adapts 2C -> 128), and the bi-GRU
    initial state s = relu(dense(spk, 128)) shared by both directions."""
    bank = [relu(conv1d_same(x, p[prefix + 'bank_%d/kernel' % k], p[prefix + 'bank_%d/bias' % k]))
            for k in range(1, K + 1)]
    y = np.concatenate(bank, -1)
    y = bn_affine(y, p[prefix + 'bank_bn/gamma'], p[prefix + 'bank_bn/beta'])
    y = maxpool2_same(y)
    y = relu(conv1d_same(y, p[prefix + 'proj1/kernel'], p[prefix + 'proj1/bias']))
    y = bn_affine(y, p[prefix + 'proj1_bn/gamma'], p[prefix + 'proj1_bn/beta'])
    y = conv1d_same(y, p[prefix + 'proj2/kernel'], p[prefix + 'proj2/bias'])
    y = bn_affine(y, p[prefix + 'proj2_bn/gamma'], p[prefix + 'proj2_bn/beta'])
    h = y + x
    for l in range(4):
        hp = prefix + 'highway_%d/' % l
        if speaker_embed is not None:
            sv = relu(dense(speaker_embed, p[hp + 'spk/kernel'], p[hp + 'spk/bias']))
            h = np.concatenate([h, np.repeat(sv[:, None, :], h.shape[1], axis=1)], -1)
        h = highway(h, p, hp)
    h0 = None
    if speaker_embed is not None:
        h0 = relu(dense(speaker_embed, p[prefix + 'gru_init/kernel'], p[prefix + 'gru_init/bias']))
    return bigru(h, p, prefix + 'bigru/', h0)


def pre_net(x, p, prefix, keep1=None, keep2=None):
    """Tacotron.pre_net (tacotron.py:38-44).  keep masks are 0/1 arrays (None => inference, no dropout)."""
    l1 = relu(dense(x, p[prefix + 'dense/kernel'], p[prefix + 'dense/bias']))
    if keep1 is not None:
        l1 = l1 * keep1 * 2
    l2 = relu(dense(l1, p[prefix + 'dense_1/kernel'], p[prefix + 'dense_1/bias']))
    if keep2 is not None:
        l2 = l2 * keep2 * 2
    return l2


# ----------------------------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------------------------
def encoder(p, text, keep1=None, keep2=None, speaker=None):
    """embedding + (speaker embedding, tacotron.py:117-124) + encoder pre_net + CBHG(K=16) (tacotron.py:111-131)."""
    emb = p['embedding'][text]
    spk = p['speaker_embed'][speaker] if (speaker is not None and 'speaker_embed' in p) else None
    pre = pre_net(emb, p, 'encoder/pre_net/', keep1, keep2)
    return cbhg(pre, p, 'encoder/cbhg/', 16, spk)


def attention_memory(p, encoded, text_length):
    """BahdanauAttention.__init__: masked values + keys (tacotron.py:48-52)."""
    B, Tt, _ = encoded.shape
    mask = (np.arange(Tt)[None, :] < np.asarray(text_length)[:, None])
    values = encoded * mask[:, :, None].astype(encoded.dtype)
    keys = values @ p['decoder/memory_layer/kernel']
    return values, keys, mask


def decoder(p, encoded, text_length, r, n_steps, mel=None, sample_mask=None, keep1=None, keep2=None):
    """create_decoder + dynamic_decode (tacotron.py:46-105, 134-138).

    mel:         (B, Td, 80r) teacher inputs, or None => InferenceHelper (zeros start, feed back outputs).
    sample_mask: (Td, B) 0/1; mask[t,b]=1 => step t+1 of row b is fed cell_output[t] instead of mel[t+1].
    keep1/keep2: (B, Td, 256)/(B, Td, 128) decoder pre-net dropout keep masks, or None.
    Returns seq2seq_output (B,Td,80r), alignments (B,Td,Tt).
    """
    B, Tt, _ = encoded.shape
    dt = encoded.dtype
    nmel = 80
    values, keys, mask = attention_memory(p, encoded, text_length)
    v = p['decoder/attention_v']
    h = [np.zeros((B, 256), dtype=dt) for _ in range(3)]
    att = np.zeros((B, 256), dtype=dt)
    outs = np.zeros((B, n_steps, nmel * r), dtype=dt)
    aligns = np.zeros((B, n_steps, Tt), dtype=dt)
    prev = mel[:, 0, :] if mel is not None else np.zeros((B, nmel * r), dtype=dt)
    for t in range(n_steps):
        k1 = keep1[:, t, :] if keep1 is not None else None
        k2 = keep2[:, t, :] if keep2 is not None else None
        pn = pre_net(prev[:, nmel * (r - 1):], p, 'decoder/pre_net/', k1, k2)
        x = dense(np.concatenate([pn, att], -1), p['decoder/in_proj/kernel'], p['decoder/in_proj/bias'])
        inp = x
        for l in range(3):
            h[l], _, _, _ = gru_cell(inp, h[l],
                                     p['decoder/gru_%d/gates/kernel' % l], p['decoder/gru_%d/gates/bias' % l],
                                     p['decoder/gru_%d/candidate/kernel' % l], p['decoder/gru_%d/candidate/bias' % l])
            inp = h[l]
        o = dense(x + h[2], p['decoder/out_proj/kernel'], p['decoder/out_proj/bias'])
        q = o @ p['decoder/query_layer/kernel']
        e = np.sum(v * np.tanh(keys + q[:, None, :]), -1)
        e = np.where(mask, e, -np.inf)
        e = e - e.max(-1, keepdims=True)
        a = np.exp(e)
        a = a / a.sum(-1, keepdims=True)
        ctx = np.einsum('bs,bsu->bu', a, values)
        att = np.concatenate([o, ctx], -1) @ p['decoder/attention_layer/kernel']
        outs[:, t, :] = o
        aligns[:, t, :] = a
        if mel is None:
            prev = o
        elif t + 1 < n_steps:
            prev = mel[:, t + 1, :]
            if sample_mask is not None:
                m = sample_mask[t].astype(bool)[:, None]
                prev = np.where(m, o, prev)
    return outs, aligns


def postnet(p, seq2seq_output, r):
    """post-process CBHG(K=8) + dense(1025) (tacotron.py:142-152)."""
    B, Td, _ = seq2seq_output.shape
    x = seq2seq_output.reshape(B, Td * r, 80)
    y = cbhg(x, p, 'post/cbhg/', 8)
    y = dense(y, p['post/dense/kernel'], p['post/dense/bias'])
    return y.reshape(B, Td, -1)


def forward(p, inputs, r, n_steps, train, masks=None):
    """Tacotron.inference (tacotron.py:107-154).  masks: dict with enc_keep1, enc_keep2, dec_keep1,
    dec_keep2, sample (any may be absent)."""
    masks = masks or {}
    enc = encoder(p, inputs['text'], masks.get('enc_keep1') if train else None,
                  masks.get('enc_keep2') if train else None, inputs.get('speaker'))
    if train:
        s2s, al = decoder(p, enc, inputs['text_length'], r, n_steps, mel=inputs['mel'],
                          sample_mask=masks.get('sample'), keep1=masks.get('dec_keep1'),
                          keep2=masks.get('dec_keep2'))
    else:
        s2s, al = decoder(p, enc, inputs['text_length'], r, n_steps)
    out = postnet(p, s2s, r)
    return s2s, out, al, enc


def loss_fn(s2s, out, mel, stft):
    """add_loss_op (tacotron.py:156-165)."""
    return np.abs(s2s - mel).sum() + np.abs(out - stft).sum()


def clip_adam_step(params, grads, m, v, step, lr, cap=5.0, b1=0.9, b2=0.999, eps=1e-8):
    """add_train_op (tacotron.py:167-185): clip_by_global_norm then TF AdamOptimizer.  `step` = t (1-based)
    of this update.  All dicts of arrays; updates params/m/v in place; returns the global norm."""
    gn = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    scale = cap / max(gn, cap) if cap > 0 else 1.0
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    for k in params:
        g = grads[k] * scale
        m[k][...] = b1 * m[k] + (1 - b1) * g
        v[k][...] = b2 * v[k] + (1 - b2) * g * g
        params[k][...] = params[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)
    return gn


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
def param_spec(vocab_size, r, num_speakers=1, speaker_dim=16):
    """Ordered list of (name, shape, init) -- the same order/layout as csrc/layout.hip's table.
    init in {'glorot', 'zeros', 'ones'}.  TF-r1.2 default initialisers (SURVEY §8a footer).
    num_speakers > 1 adds the speaker table (tacotron.py:117-124) and the encoder CBHG speaker sites (ops.py:101-115)."""
    spec = [('embedding', (vocab_size, 256), 'glorot')]
    multi = num_speakers > 1
    if multi:
        spec.append(('speaker_embed', (num_speakers, speaker_dim), 'glorot'))

    def dense_(name, i, o, bias=True):
        spec.append((name + '/kernel', (i, o), 'glorot'))
        if bias:
            spec.append((name + '/bias', (o,), 'zeros'))

    def gru_(name, cin, h):
        spec.append((name + '/gates/kernel', (cin + h, 2 * h), 'glorot'))
        spec.append((name + '/gates/bias', (2 * h,), 'ones'))
        spec.append((name + '/candidate/kernel', (cin + h, h), 'glorot'))
        spec.append((name + '/candidate/bias', (h,), 'zeros'))

    def cbhg_(prefix, K, cin, c1, c2, spk=False):
        for k in range(1, K + 1):
            spec.append((prefix + 'bank_%d/kernel' % k, (k, cin, 128), 'glorot'))
            spec.append((prefix + 'bank_%d/bias' % k, (128,), 'zeros'))
        spec.append((prefix + 'bank_bn/gamma', (K * 128,), 'ones'))
        spec.append((prefix + 'bank_bn/beta', (K * 128,), 'zeros'))
        spec.append((prefix + 'proj1/kernel', (3, K * 128, c1), 'glorot'))
        spec.append((prefix + 'proj1/bias', (c1,), 'zeros'))
        spec.append((prefix + 'proj1_bn/gamma', (c1,), 'ones'))
        spec.append((prefix + 'proj1_bn/beta', (c1,), 'zeros'))
        spec.append((prefix + 'proj2/kernel', (3, c1, c2), 'glorot'))
        spec.append((prefix + 'proj2/bias', (c2,), 'zeros'))
        spec.append((prefix + 'proj2_bn/gamma', (c2,), 'ones'))
        spec.append((prefix + 'proj2_bn/beta', (c2,), 'zeros'))
        for l in range(4):
            hp = prefix + 'highway_%d/' % l
            if spk:
                dense_(hp + 'spk', speaker_dim, 128)
                dense_(hp + 'adapt', 256, 128)
            elif l == 0 and c2 != 128:
                dense_(hp + 'adapt', c2, 128)
            dense_(hp + 'T', 128, 128)
            dense_(hp + 'H', 128, 128)
        if spk:
            dense_(prefix + 'gru_init', speaker_dim, 128)
        gru_(prefix + 'bigru/fw', 128, 128)
        gru_(prefix + 'bigru/bw', 128, 128)

    dense_('encoder/pre_net/dense', 256, 256)
    dense_('encoder/pre_net/dense_1', 256, 128)
    cbhg_('encoder/cbhg/', 16, 128, 128, 128, spk=multi)
    dense_('decoder/memory_layer', 256, 256, bias=False)
    dense_('decoder/pre_net/dense', 80, 256)
    dense_('decoder/pre_net/dense_1', 256, 128)
    dense_('decoder/in_proj', 384, 256)
    for l in range(3):
        gru_('decoder/gru_%d' % l, 256, 256)
    dense_('decoder/out_proj', 256, 80 * r)
    dense_('decoder/query_layer', 80 * r, 256, bias=False)
    spec.append(('decoder/attention_v', (256,), 'glorot'))
    dense_('decoder/attention_layer', 80 * r + 256, 256, bias=False)
    cbhg_('post/cbhg/', 8, 80, 256, 80)
    dense_('post/dense', 256, 1025)
    return spec


def glorot_limit(shape):
    """TF glorot_uniform: fan_in/fan_out with receptive field for conv kernels; 1-D -> fan_in=fan_out=n."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    return np.sqrt(6.0 / (fan_in + fan_out))


def init_params(vocab_size, r, seed=0, dtype=np.float64, perturb=0.0, num_speakers=1):
    """Seeded TF-default initialisation.  `perturb` > 0 additionally jitters biases / BN affine so that
    fixtures exercise them (all-zero biases would hide indexing bugs)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, init in param_spec(vocab_size, r, num_speakers):
        if init == 'glorot':
            lim = glorot_limit(shape)
            a = rng.uniform(-lim, lim, size=shape)
        elif init == 'zeros':
            a = np.zeros(shape)
        else:
            a = np.ones(shape)
        if perturb > 0 and init != 'glorot':
            a = a + rng.uniform(-perturb, perturb, size=shape)
        p[name] = a.astype(dtype)
    return p


def flatten_params(p, vocab_size, r, dtype=np.float32, num_speakers=1):
    return np.concatenate([p[n].reshape(-1) for n, _, _ in param_spec(vocab_size, r, num_speakers)]).astype(dtype)


def unflatten_params(flat, vocab_size, r, num_speakers=1):
    p, o = {}, 0
    for n, shape, _ in param_spec(vocab_size, r, num_speakers):
        sz = int(np.prod(shape))
        p[n] = flat[o:o + sz].reshape(shape)
        o += sz
    assert o == flat.size
    return p
