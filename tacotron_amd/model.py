"""Tacotron -- host-side mirror of the reference's model object (models/tacotron.py:187-195).

Same constructor contract `Tacotron(config, inputs, train)` and the same public attributes
(`.seq2seq_output .output .alignments .loss .global_step .lr .config`); TF's `sess.run(train_op)` becomes
`.step(lr)` and `sess.run([output, alignments])` becomes `.run()`.  All arithmetic is in libtaco_hip.so.
"""
from __future__ import annotations

import torch

from . import lib
from .params import ParamBuffer


class Tacotron(object):
    def __init__(self, config, inputs, train=True, device='cuda', params=None, seed=0, reducer=None):
        config.validate()
        self.config = config
        self.train = train
        self.device = torch.device(device)
        self.reducer = reducer            # tacotron_amd.dist.GradReducer or None
        self.lr = config.init_lr
        self.global_step = 0
        self.merged = None                # tf.summary plumbing is out of scope
        B, Tt = inputs['text'].shape
        if train:
            Td = inputs['mel'].shape[1]
        else:
            Td = config.max_decode_iter
        self.shape = lib.make_shape(B, Tt, Td, config.r, config.vocab_size, config.num_speakers)
        if params is None:
            params = ParamBuffer(self.shape, self.device).init_(seed)
        self.params = params
        R80, F2 = config.mel_features * config.r, config.fft_size * config.r
        dev = self.device
        self.seq2seq_output = torch.empty(B, Td, R80, device=dev)
        self.output = torch.empty(B, Td, F2, device=dev)
        self.alignments = torch.empty(B, Td, Tt, device=dev)
        self._loss = torch.zeros(3, device=dev)
        self.workspace = torch.empty(lib.workspace_bytes(self.shape, train) // 4, dtype=torch.float32, device=dev)
        self.masks = None
        self._err_off = None
        if train:
            n = params.numel
            self.grads = torch.zeros(n, device=dev)
            self.adam_m = torch.zeros(n, device=dev)
            self.adam_v = torch.zeros(n, device=dev)
            self._scratch = torch.zeros(8, device=dev)
            self._gnorm = torch.zeros(1, device=dev)
            # the five mask tensors are views of ONE byte buffer (16-byte aligned pieces), so that neighbours with the same
            # Bernoulli parameter are filled by a single launch (all five, with the reference's default rates)
            shapes = (('enc_keep1', (B, Tt, 256)), ('enc_keep2', (B, Tt, 128)), ('dec_keep1', (B, Td, 256)),
                      ('dec_keep2', (B, Td, 128)), ('sample', (Td, B)))
            sizes = [((s[0] * s[1] * (s[2] if len(s) > 2 else 1)) + 15) // 16 * 16 for _, s in shapes]
            self._mask_flat = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
            self._mask_buf, self._mask_span, off = {}, {}, 0
            for (k, s), sz in zip(shapes, sizes):
                n = s[0] * s[1] * (s[2] if len(s) > 2 else 1)
                self._mask_buf[k] = self._mask_flat[off:off + n].view(*s)
                self._mask_span[k] = (off, sz)
                off += sz
            self._seed = seed * 1000003 + 17
        self.set_inputs(inputs)

    # -- inputs -----------------------------------------------------------------------------------------
    def set_inputs(self, inputs):
        dev = self.device
        self.inputs = {
            'text': inputs['text'].to(dev, torch.int32).contiguous(),
            'text_length': inputs['text_length'].to(dev, torch.int32).contiguous(),
        }
        self.speaker = None
        if self.config.num_speakers > 1:   # tacotron.py:117-124
            self.speaker = inputs['speaker'].to(dev, torch.int32).contiguous()
        if self.train:
            self.inputs['mel'] = inputs['mel'].to(dev, torch.float32).contiguous()
            self.inputs['stft'] = inputs['stft'].to(dev, torch.float32).contiguous()

    def draw_masks(self):
        """Dropout keep masks (keep prob 1-rate) and the scheduled-sampling Bernoulli mask (tacotron.py:41-43,84-85)."""
        c = self.config
        m = {}
        self._seed += 5
        want = [('enc_keep1', 1.0 - c.char_dropout_prob if c.char_dropout_prob else None),
                ('enc_keep2', 1.0 - c.char_dropout_prob if c.char_dropout_prob else None),
                ('dec_keep1', 1.0 - c.audio_dropout_prob if c.audio_dropout_prob else None),
                ('dec_keep2', 1.0 - c.audio_dropout_prob if c.audio_dropout_prob else None),
                ('sample', c.scheduled_sample if c.scheduled_sample else None)]
        i = 0
        while i < len(want):
            k, p = want[i]
            if p is None:
                i += 1
                continue
            j = i
            while j + 1 < len(want) and want[j + 1][1] == p:   # run of neighbours with the same parameter: one launch
                j += 1
            off = self._mask_span[k][0]
            end = self._mask_span[want[j][0]][0] + self._mask_span[want[j][0]][1]
            lib.fill_bernoulli(self._mask_flat[off:end], p, self._seed + i)
            for q in range(i, j + 1):
                m[want[q][0]] = self._mask_buf[want[q][0]]
            i = j + 1
        return m

    # -- train ------------------------------------------------------------------------------------------
    def forward(self, masks=None):
        i = self.inputs
        self.masks = masks
        lib.forward(self.shape, self.params.flat, i['text'], i['text_length'], i['mel'], i['stft'], masks,
                    self.seq2seq_output, self.output, self.alignments, self._loss, self.workspace, self.speaker)

    def backward(self):
        i = self.inputs
        lib.backward(self.shape, self.params.flat, i['text'], i['text_length'], self.seq2seq_output, self.alignments,
                     self.masks, self.grads, self.workspace, self.speaker)

    def apply_gradients(self, lr):
        self.global_step += 1
        lib.clip_adam_step(self.params.flat, self.grads, self.adam_m, self.adam_v, lr, self.config.cap_grads,
                           self.global_step, self._scratch, self._gnorm)

    def step(self, lr=None, masks='draw'):
        """One `sess.run(train_op)`: forward + backward (+ gradient all-reduce) + clip + Adam.  Everything is
        enqueued on the current stream; nothing is copied to the host."""
        if masks == 'draw':
            masks = self.draw_masks()
        self.forward(masks)
        self.backward()
        if self.reducer is not None:
            self.reducer.all_reduce(self.grads, self._loss)
        self.apply_gradients(self.lr if lr is None else lr)

    def check(self):
        """Raises TacoError if a decoder kernel reported a timed-out cluster exchange (the kernels never hang: every spin is
        bounded and a time-out sets an error word in the workspace).  This is a host synchronisation -- the drivers call
        it where they already synchronise (loss logging, after inference), not every step."""
        if self._err_off is None:
            self._err_off = [o for name, o, s, d in lib.workspace_table(self.shape, self.train) if name == 'dec.err'][0]
        flags = self.workspace[self._err_off:self._err_off + 2].view(torch.int32).tolist()
        if flags[0] or flags[1]:
            raise lib.TacoError('decoder cluster exchange timed out (forward=%d, backward=%d)' % (flags[0], flags[1]))

    @property
    def loss(self):
        return self._loss[0]

    @property
    def global_gradient_norm(self):
        return self._gnorm[0]

    # -- inference --------------------------------------------------------------------------------------
    def run(self):
        """`sess.run([model.output, model.alignments])` for train=False (test.py:52-56)."""
        i = self.inputs
        lib.infer(self.shape, self.params.flat, i['text'], i['text_length'], self.seq2seq_output, self.output,
                  self.alignments, self.workspace, self.speaker)
        return self.output, self.alignments

    # -- checkpoint (train.py:47,85-90: weights + Adam slots + global_step) --------------------------------
    def state_dict(self):
        d = {'params': self.params.flat.detach().cpu(), 'global_step': self.global_step,
             'shape': (self.shape.r, self.shape.V)}
        if self.train:
            d['adam_m'] = self.adam_m.cpu()
            d['adam_v'] = self.adam_v.cpu()
        return d

    def load_state_dict(self, d):
        self.params.flat.copy_(d['params'])
        self.global_step = int(d.get('global_step', 0))
        if self.train and 'adam_m' in d:
            self.adam_m.copy_(d['adam_m'])
            self.adam_v.copy_(d['adam_v'])
