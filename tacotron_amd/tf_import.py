"""Import of the reference's TF-1.2 checkpoint variables (download_weights.sh:4-7, train.py:31-33,47-58).

The flat parameter buffer of libtaco_hip.so keeps TF's variable ORDER and LAYOUTS (dense (in,out), conv1d (k,Cin,Cout),
GRUCell gates (Cin+H, 2H) r-then-u), so importing a checkpoint is a rename plus one copy per tensor -- no transposes.
This module holds the rename table as code, derived by reading the reference graph (models/tacotron.py:35-154,
models/ops.py:27-132) against TensorFlow 1.2's naming rules:
  * tf.layers.* layers without a name get `<class>`, `<class>_1`, ... per enclosing variable scope, in creation order
    (conv bank k = 1..16 -> conv1d .. conv1d_15, projections conv1d_16 / conv1d_17, BN batch_normalization[_1,_2]);
  * highway(): optional input adapter first, then T, then H; with speakers the per-layer speaker dense comes before them;
  * GRUCell r1.2: `gates/{kernel,bias}`, `candidate/{kernel,bias}`; bidirectional_dynamic_rnn: `bidirectional_rnn/{fw,bw}/gru_cell`;
  * dynamic_decode opens a second `decoder` scope; AttentionWrapper -> `attention_wrapper`, BahdanauAttention's layers
    `memory_layer` (built at construction, outside the loop scope), `bahdanau_attention/query_layer`, `.../attention_v`,
    `attention_layer`; the wrapped cell: `output_projection_wrapper/{kernel,bias}` > `input_projection_wrapper/{kernel,bias}` >
    `multi_rnn_cell/cell_<l>/gru_cell/...`; the decoder pre_net is built inside the wrapper's call (cell_input_fn).
STATUS: the table is UNVERIFIED against a real checkpoint (the pretrained weights are a Dropbox download this environment
cannot reach, and TensorFlow is not installable here).  `check_names()` reports every name that does not line up, so the first
run next to the real file either confirms the table or names exactly what to fix."""
from __future__ import annotations

import numpy as np
import torch

from . import lib
from .params import ParamBuffer


def _cbhg(prefix_ours, prefix_tf, K, has_adapt0, spk):
    m = {}
    for k in range(1, K + 1):
        tf = 'conv1d' if k == 1 else 'conv1d_%d' % (k - 1)
        m['%sbank_%d/kernel' % (prefix_ours, k)] = '%s%s/kernel' % (prefix_tf, tf)
        m['%sbank_%d/bias' % (prefix_ours, k)] = '%s%s/bias' % (prefix_tf, tf)
    m[prefix_ours + 'bank_bn/gamma'] = prefix_tf + 'batch_normalization/gamma'
    m[prefix_ours + 'bank_bn/beta'] = prefix_tf + 'batch_normalization/beta'
    for i, name in enumerate(('proj1', 'proj2')):
        m['%s%s/kernel' % (prefix_ours, name)] = '%sconv1d_%d/kernel' % (prefix_tf, K + i)
        m['%s%s/bias' % (prefix_ours, name)] = '%sconv1d_%d/bias' % (prefix_tf, K + i)
        m['%s%s_bn/gamma' % (prefix_ours, name)] = '%sbatch_normalization_%d/gamma' % (prefix_tf, i + 1)
        m['%s%s_bn/beta' % (prefix_ours, name)] = '%sbatch_normalization_%d/beta' % (prefix_tf, i + 1)
    for l in range(4):
        ours, tf = '%shighway_%d/' % (prefix_ours, l), '%shighway_%d/' % (prefix_tf, l)
        seq = []
        if spk:
            seq.append('spk')
        if spk or (l == 0 and has_adapt0):
            seq.append('adapt')
        seq += ['T', 'H']
        for i, part in enumerate(seq):
            d = 'dense' if i == 0 else 'dense_%d' % i
            m['%s%s/kernel' % (ours, part)] = '%s%s/kernel' % (tf, d)
            m['%s%s/bias' % (ours, part)] = '%s%s/bias' % (tf, d)
    if spk:
        m[prefix_ours + 'gru_init/kernel'] = prefix_tf + 'dense/kernel'
        m[prefix_ours + 'gru_init/bias'] = prefix_tf + 'dense/bias'
    for d in ('fw', 'bw'):
        for part in ('gates', 'candidate'):
            for leaf in ('kernel', 'bias'):
                m['%sbigru/%s/%s/%s' % (prefix_ours, d, part, leaf)] = '%sbidirectional_rnn/%s/gru_cell/%s/%s' % (prefix_tf, d, part, leaf)
    return m


def tf_name_map(num_speakers=1):
    """{libtaco parameter name: TF-1.2 variable name (without ':0')}."""
    spk = num_speakers > 1
    m = {'embedding': 'embedding/embedding'}
    if spk:
        m['speaker_embed'] = 'speaker/speaker_embed'
    for i, d in enumerate(('dense', 'dense_1')):
        for leaf in ('kernel', 'bias'):
            m['encoder/pre_net/%s/%s' % (d, leaf)] = 'encoder/pre_net/%s/%s' % (d, leaf)
    m.update(_cbhg('encoder/cbhg/', 'encoder/cbhg/', 16, False, spk))
    m['decoder/memory_layer/kernel'] = 'decoder/memory_layer/kernel'
    aw = 'decoder/decoder/attention_wrapper/'
    for d in ('dense', 'dense_1'):
        for leaf in ('kernel', 'bias'):
            m['decoder/pre_net/%s/%s' % (d, leaf)] = '%spre_net/%s/%s' % (aw, d, leaf)
    opw = aw + 'output_projection_wrapper/'
    ipw = opw + 'input_projection_wrapper/'
    for leaf in ('kernel', 'bias'):
        m['decoder/in_proj/' + leaf] = ipw + leaf
        m['decoder/out_proj/' + leaf] = opw + leaf
    for l in range(3):
        for part in ('gates', 'candidate'):
            for leaf in ('kernel', 'bias'):
                m['decoder/gru_%d/%s/%s' % (l, part, leaf)] = '%smulti_rnn_cell/cell_%d/gru_cell/%s/%s' % (ipw, l, part, leaf)
    m['decoder/query_layer/kernel'] = aw + 'bahdanau_attention/query_layer/kernel'
    m['decoder/attention_v'] = aw + 'bahdanau_attention/attention_v'
    m['decoder/attention_layer/kernel'] = aw + 'attention_layer/kernel'
    m.update(_cbhg('post/cbhg/', 'post-process/cbhg/', 8, True, False))
    m['post/dense/kernel'] = 'post-process/dense/kernel'
    m['post/dense/bias'] = 'post-process/dense/bias'
    return m


IGNORED_SUFFIXES = ('/moving_mean', '/moving_variance')   # BN runs in inference mode with the initial statistics


def check_names(checkpoint_names, shape):
    """Lines up a checkpoint's variable names with the table.  Returns (missing, unexpected): table entries absent from the
    checkpoint, and checkpoint variables nothing maps to (optimizer slots, BN moving statistics and bookkeeping excluded)."""
    names = {n[:-2] if n.endswith(':0') else n for n in checkpoint_names}
    ours = {n for n, _, _, _ in lib.param_table(shape)}
    table = tf_name_map(max(1, shape.S))
    assert set(table) == ours, sorted(set(table) ^ ours)
    wanted = set(table.values())
    missing = sorted(wanted - names)
    extra = sorted(n for n in names - wanted
                   if not n.endswith(IGNORED_SUFFIXES) and not n.endswith(('/Adam', '/Adam_1'))
                   and n not in ('global_step', 'stft_mean', 'stft_std', 'beta1_power', 'beta2_power'))
    return missing, extra


def import_tf_variables(variables, shape, device='cpu'):
    """variables: {TF variable name: array} (e.g. from tf.train.load_checkpoint on a machine that has TF, saved as .npz).
    Returns a state dict for `Tacotron.load_state_dict`: params, Adam slots when present, global_step, stft_mean / stft_std."""
    v = {(k[:-2] if k.endswith(':0') else k): np.asarray(a) for k, a in variables.items()}
    missing, _ = check_names(v.keys(), shape)
    if missing:
        raise KeyError('checkpoint lacks %d variables the graph needs, e.g. %s' % (len(missing), missing[:4]))
    table = tf_name_map(max(1, shape.S))
    pb = ParamBuffer(shape, device)
    slots = {}
    for suffix, key in (('', 'params'), ('/Adam', 'adam_m'), ('/Adam_1', 'adam_v')):
        have = all((table[n] + suffix) in v for n in pb.names())
        if not have:
            continue
        d = {}
        for n, off, size, dims in pb.table:
            a = v[table[n] + suffix]
            if tuple(a.shape) != tuple(dims):
                raise ValueError('%s: checkpoint shape %s, expected %s' % (table[n], a.shape, dims))
            d[n] = a
        slots[key] = ParamBuffer(shape, device).load_dict_(d).flat.clone()
    out = {'params': slots['params'], 'shape': (shape.r, shape.V), 'num_speakers': max(1, shape.S),
           'global_step': int(v['global_step']) if 'global_step' in v else 0, 'taco_version': lib.version()}
    for k in ('adam_m', 'adam_v'):
        if k in slots:
            out[k] = slots[k]
    for k in ('stft_mean', 'stft_std'):   # train.py:31-33
        if k in v:
            out[k] = torch.as_tensor(v[k], dtype=torch.float32)
    return out
