"""Config -- the reference's hyper-parameter object (models/tacotron.py:12-33) with the same field names and
defaults, plus the audio constants it derives `max_decode_iter` from (audio.py:10-17)."""
from __future__ import annotations

# audio.py:10-17
n_fft = 2048
win_length = 1200
hop_length = int(win_length / 4)
maximum_audio_length = 108000
r = 2

# data_input.py:15-18
BATCH_SIZE = 32
SHUFFLE_BUFFER_SIZE = 10000
MAX_TEXT_LEN = 140

# train.py:13-14
SAVE_EVERY = 5000
RESTORE_FROM = None


class Config(object):
    """Same attribute names as the reference; mutated at runtime by train()/test() exactly as the reference
    drivers do (`config.r`, `config.vocab_size`, `config.data_path`, `config.save_path`, `config.restore`)."""
    max_decode_iter = maximum_audio_length // (r * hop_length)
    attention_units = 256
    decoder_units = 256
    mel_features = 80
    embed_dim = 256
    fft_size = 1025

    char_dropout_prob = 0.5
    audio_dropout_prob = 0.5

    num_speakers = 1
    speaker_embed_dim = 16

    scheduled_sample = 0.5

    cap_grads = 5

    init_lr = 0.0005
    annealing_rate = 1

    batch_size = 32

    # runtime-added in the reference (train.py:20-22,118-123); given defaults here so a bare Config() is usable
    r = r
    vocab_size = 60
    data_path = 'data/nancy/'
    save_path = 'nancy/tacotron'
    restore = False

    def validate(self):
        """The HIP kernels compile the reference's layer widths in (include/taco_hip.h); refuse anything else loudly."""
        fixed = dict(attention_units=256, decoder_units=256, mel_features=80, embed_dim=256, fft_size=1025)
        for k, v in fixed.items():
            if getattr(self, k) != v:
                raise ValueError('Config.%s=%r is not supported by libtaco_hip (compiled for %r)' % (k, getattr(self, k), v))
        if self.num_speakers < 1:
            raise ValueError('Config.num_speakers must be >= 1')
        if self.num_speakers > 1 and self.speaker_embed_dim != 16:
            raise ValueError('Config.speaker_embed_dim=%r is not supported by libtaco_hip (compiled for 16)' % self.speaker_embed_dim)
        if not (1 <= self.r <= 5):
            raise ValueError('Config.r must be in 1..5')
        for k in ('char_dropout_prob', 'audio_dropout_prob'):
            if getattr(self, k) not in (0, 0.0, 0.5):
                raise ValueError('Config.%s must be 0 or 0.5 (dropout scale 2 is compiled in)' % k)
