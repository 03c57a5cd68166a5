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
    ESCALATE_WINDOW = 2000   # steps: two exchange time-outs closer than this escalate the decoder mode (check())

    def __init__(self, config, inputs, train=True, device='cuda', params=None, seed=0, reducer=None):
        config.validate()
        self._last_timeout_step = None
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
        # the decoder kernels' two error words (sticky: cleared here and by check(), never by the library)
        eoff = [o for name, o, s, d in lib.workspace_table(self.shape, train) if name == 'dec.err'][0]
        self._err = self.workspace[eoff:eoff + 2].view(torch.int32)
        self._census = self.workspace[eoff + 4:eoff + 12].view(torch.int32)
        lib.clear_error(self.shape, train, self.workspace)
        self.stft_mean = self.stft_std = None   # train.py:31-33: normalisation statistics travel with the checkpoint
        if train:
            n = params.numel
            self.grads = torch.zeros(n, device=dev)
            self.adam_m = torch.zeros(n, device=dev)
            self.adam_v = torch.zeros(n, device=dev)
            self._scratch = torch.zeros(256, device=dev)
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
            # per-rank dropout / scheduled-sampling streams: replicas share the parameters, not the masks
            rank = reducer.rank if reducer is not None else 0
            self._seed = seed * 1000003 + 17 + rank * 7919
        self.set_inputs(inputs)

    # -- inputs -----------------------------------------------------------------------------------------
    def set_inputs(self, inputs):
        dev = self.device
        sh = self.shape
        want = {'text': (sh.B, sh.Tt), 'text_length': (sh.B,)}
        if self.train:
            want.update(mel=(sh.B, sh.Td, 80 * sh.r), stft=(sh.B, sh.Td, 1025 * sh.r))
        if self.config.num_speakers > 1:
            want['speaker'] = (sh.B,)
        for k, shp in want.items():
            if k not in inputs:
                raise lib.TacoError('inputs[%r] is missing' % k)
            if tuple(inputs[k].shape) != shp:   # the kernels index with self.shape: a differently shaped batch would read past the tensors
                raise lib.TacoError('inputs[%r] has shape %s, this model was built for %s' % (k, tuple(inputs[k].shape), shp))
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
        """Guarded: if a decoder kernel flagged a timed-out exchange (on ANY rank -- the error words are MAX-reduced with
        the gradients) the update is skipped on the device and `global_gradient_norm` reads -1; check() raises.
        `global_step` counts ATTEMPTED updates (it is advanced on the host before the device decides): after a skipped update
        the Adam bias-correction step, the annealing cadence and the checkpoint numbering run ahead of the updates actually
        applied by the number of skipped steps -- which check() reports at the next synchronisation point, where the drivers
        stop."""
        self.global_step += 1
        lib.clip_adam_step(self.params.flat, self.grads, self.adam_m, self.adam_v, lr, self.config.cap_grads,
                           self.global_step, self._scratch, self._gnorm, self._err)

    def step(self, lr=None, masks='draw'):
        """One `sess.run(train_op)`: forward + backward (+ gradient all-reduce, overlapped with the backward pass) +
        clip + Adam.  Everything is enqueued on the current stream; nothing is copied to the host."""
        if masks == 'draw':
            masks = self.draw_masks()
        self.forward(masks)
        self.backward()
        if self.reducer is not None:
            self.reducer.reduce_after_backward(self)
        self.apply_gradients(self.lr if lr is None else lr)

    def check(self):
        """Raises TacoError if a decoder kernel reported a timed-out cluster exchange (the kernels never hang: every spin is
        bounded and a time-out sets a STICKY error word in the workspace, which also makes the Adam update skip itself).
        This is a host synchronisation -- the drivers call it where they already synchronise (loss logging, after
        inference).  The words are cleared here, after the error has been turned into an exception."""
        flags = self._err.tolist()
        if flags[0] or flags[1]:
            lib.clear_error(self.shape, self.train, self.workspace)
            # self-heal: a SECOND time-out within ESCALATE_WINDOW steps of the previous one moves this PROCESS to the next more
            # conservative decoder mode (include/taco_hip.h taco_decoder_mode: XCD-local exchange -> agent-scope exchange ->
            # decoder.hip, ~2 x slower per decoder step).  A single time-out -- e.g. another tenant briefly holding CUs -- only
            # costs the updates skipped while the flag was set: it must not degrade the rest of a multi-day run.
            mode = lib.decoder_mode()
            prev = self._last_timeout_step
            self._last_timeout_step = self.global_step
            repeat = prev is not None and self.global_step - prev <= self.ESCALATE_WINDOW
            # error word 2 = the placement rendezvous of a decoder3 launch found a cluster NOT CO-RESIDENT (a peer workgroup was not
            # dispatched within 50 ms: something else holds CUs the persistent kernel needs).  That is a diagnosis, not a transient:
            # both decoder3 modes need all 256 workgroups resident and every such launch costs 50 ms, so the process goes straight
            # to decoder.hip (mode 2), whose launch geometry is sized from the occupancy query.
            not_resident = 2 in (flags[0], flags[1]) and mode < 2
            escalate = (repeat or not_resident) and mode < 2
            new_mode = 2 if not_resident else mode + 1
            what = ('decoder cluster NOT CO-RESIDENT (error word 2: a peer workgroup was not dispatched within 50 ms -- another kernel '
                    'holds CUs the persistent decoder needs)' if not_resident else 'decoder cluster exchange timed out')
            e = lib.TacoError('%s (forward=%d, backward=%d) in decoder mode %d at step %d; parameter '
                              'updates were skipped while the flag was set%s' %
                              (what, flags[0], flags[1], mode, self.global_step,
                               ('; switched to decoder mode %d' % new_mode if not_resident else
                                '; second time-out within %d steps: switched to decoder mode %d' % (self.ESCALATE_WINDOW, new_mode))
                               if escalate else ('' if mode >= 2 else '; mode kept (first time-out in this window)')))
            e.recoverable = mode < 2
            e.not_resident = not_resident
            if escalate:
                lib.decoder_mode(new_mode)
            raise e

    def placement_census(self):
        """Diagnostic, accumulated over the decoder-forward launches since the last clear_error(); host synchronisation.
        decoder3.hip (cluster width 32): [workgroups whose cluster exchanges through its XCD's L2, workgroups of clusters that
        straddle XCDs (or TACO_DEC_V3_AGENT=1) and exchange at agent scope, 0, ...] -- the second number should be 0.
        decoder.hip (cluster width <= 16): histogram of (blockIdx - XCD id) mod 8; one non-zero bin = the fast placement."""
        return self._census.tolist()

    @property
    def loss(self):
        return self._loss[0]

    @property
    def loss_terms(self):
        """(seq2seq_loss, output_loss) -- the two sums of add_loss_op (tacotron.py:158-160) behind `loss`; device tensors."""
        return self._loss[1], self._loss[2]

    @property
    def decoder_mode(self):
        """The process-wide decoder mode in use (0 XCD-local exchange, 1 agent-scope exchange, 2 decoder.hip); see check()."""
        return lib.decoder_mode()

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
        """Weights + Adam slots + global_step + the spectrogram normalisation statistics (the reference keeps stft_mean /
        stft_std as checkpointed variables, train.py:31-33, and test.py:27-28,64 de-normalises with them)."""
        d = {'params': self.params.flat.detach().cpu(), 'global_step': self.global_step,
             'shape': (self.shape.r, self.shape.V), 'num_speakers': max(1, self.shape.S), 'taco_version': lib.version(),
             'decoder_mode': lib.decoder_mode()}
        if self.stft_mean is not None:
            d['stft_mean'] = torch.as_tensor(self.stft_mean, dtype=torch.float32).cpu()
            d['stft_std'] = torch.as_tensor(self.stft_std, dtype=torch.float32).cpu()
        if self.train:
            d['adam_m'] = self.adam_m.cpu()
            d['adam_v'] = self.adam_v.cpu()
        return d

    def load_state_dict(self, d):
        if tuple(d.get('shape', (self.shape.r, self.shape.V))) != (self.shape.r, self.shape.V) or \
                int(d.get('num_speakers', max(1, self.shape.S))) != max(1, self.shape.S) or \
                d['params'].numel() != self.params.numel:
            raise lib.TacoError('checkpoint was written for (r, V)=%s, %s speaker(s), %d parameters; this model has (r, V)=%s, '
                                '%d speaker(s), %d parameters' % (d.get('shape'), d.get('num_speakers', '?'), d['params'].numel(),
                                                                  (self.shape.r, self.shape.V), max(1, self.shape.S), self.params.numel))
        self.params.flat.copy_(d['params'])
        self.global_step = int(d.get('global_step', 0))
        if 'stft_mean' in d:
            self.stft_mean, self.stft_std = d['stft_mean'], d['stft_std']
        if self.train and 'adam_m' in d:
            self.adam_m.copy_(d['adam_m'])
            self.adam_v.copy_(d['adam_v'])
