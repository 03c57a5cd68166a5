"""ctypes binding of libtaco_hip.so (include/taco_hip.h).

PyTorch is used only as the device allocator / stream owner: every call passes raw device pointers
(`tensor.data_ptr()`) and the current HIP stream.  There is NO CPU fallback: if the shared library is missing
the import of this module raises, and every entry point raises `TacoError` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  One train step uses the caller's
# stream, the library's side stream, the weight-gradient stream and -- data parallel -- a communication stream plus RCCL's own:
# with 4 queues two of them share a queue and one's kernels sit behind a millisecond of the other's already-enqueued launches
# (measured in round 4: the post-net segment's collective finished 1.47 ms after it became eligible, 30 us with 8 queues).
# The variable is read when the HIP runtime initialises, i.e. at the first device call -- set it before anything touches the GPU.
from . import QUEUES_SET_BY_USER as _QUEUES_SET_BY_USER, TORCH_IMPORTED_FIRST as _TORCH_IMPORTED_FIRST   # (package __init__ applied the default)

import torch  # noqa: E402

# If the process may have touched the GPU before importing this package (and did not set the variable itself) the runtime is up
# with its default of 4 queues: the default came too late, and high-priority communication streams would then cost 1.3 ms per
# step instead of saving one (dist._comm_priority asks here).  "May have": the runtime reads the variable at the FIRST HIP API call
# -- hipGetDeviceCount behind torch.cuda.is_available() / device_count() counts, and leaves no trace in torch
# (torch.cuda.is_initialized() stays False) -- so every import of torch ahead of this package is treated as late.
# Launchers that import torch first set the variable themselves before doing so (bench.py, tests/conftest.py, the drivers).
HW_QUEUES_LATE = (not _QUEUES_SET_BY_USER) and (_TORCH_IMPORTED_FIRST or (torch.cuda.is_available() and torch.cuda.is_initialized()))


def effective_hw_queues() -> int:
    """Hardware queues the HIP runtime of this process multiplexes its streams onto, as far as the host can know."""
    if HW_QUEUES_LATE:
        return 4
    try:
        return int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        return 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TACO_LIB') or os.path.join(_HERE, 'libtaco_hip.so')   # TACO_LIB: an alternative build (tuning A/B)


class TacoError(RuntimeError):
    pass


class TacoShape(C.Structure):
    _fields_ = [('B', C.c_int32), ('Tt', C.c_int32), ('Td', C.c_int32), ('r', C.c_int32), ('V', C.c_int32),
                ('S', C.c_int32)]

    def __repr__(self):
        return 'TacoShape(B=%d, Tt=%d, Td=%d, r=%d, V=%d, S=%d)' % (self.B, self.Tt, self.Td, self.r, self.V, self.S)


class TacoTensorInfo(C.Structure):
    _fields_ = [('name', C.c_char * 64), ('offset', C.c_int64), ('size', C.c_int64), ('ndim', C.c_int32),
                ('dims', C.c_int32 * 4)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        'tacotron_amd: %s not found -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
        '(or tacotron_amd/csrc/build.sh).  There is no CPU fallback.' % LIB_PATH)

_lib = C.CDLL(LIB_PATH)

_P = C.c_void_p
_I = C.c_int
_SH = C.POINTER(TacoShape)

EXPORTS = {
    'taco_version': (C.c_int, []),
    'taco_last_error_string': (C.c_char_p, []),
    'taco_param_count': (C.c_int64, [_SH]),
    'taco_param_table': (C.c_int, [_SH, C.POINTER(TacoTensorInfo), _I]),
    'taco_workspace_bytes': (C.c_int64, [_SH, _I]),
    'taco_workspace_table': (C.c_int, [_SH, _I, C.POINTER(TacoTensorInfo), _I]),
    'taco_conv_gemm': (C.c_int, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'taco_debug_gemm2_window': (C.c_int, [_I, _I]),
    'taco_debug_weight_image': (C.c_int64, [_P, _I, _I, _I, _I, _P, C.c_int64, _P]),
    'taco_debug_conv_gemm_nld': (C.c_int, [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    'taco_debug_conv_gemm_ksplit': (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, C.c_int64, _P]),
    'taco_gemm_tn': (C.c_int, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'taco_debug_gemm_naive': (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'taco_bigru_fwd': (C.c_int, [_P] * 12 + [_I, _I, _P]),
    'taco_forward': (C.c_int, [_SH] + [_P] * 17),
    'taco_backward': (C.c_int, [_SH] + [_P] * 14),
    'taco_infer': (C.c_int, [_SH] + [_P] * 9),
    'taco_clip_adam_step': (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_int64, _P, _P, _P]),
    'taco_clip_adam_step_guarded': (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_int64, _P, _P, _P, _P]),
    'taco_clear_error': (C.c_int, [_SH, _I, _P, _P]),
    'taco_grad_segments': (C.c_int, [_SH, C.POINTER(C.c_int64)]),
    'taco_wait_grad_segment': (C.c_int, [_I, _P]),
    'taco_decoder_mode': (C.c_int, [_I]),
    'taco_debug_spin': (C.c_int, [_I, _I, _I, _I, _P]),
    'taco_debug_clock_probe': (C.c_int, [_P, _I, _P]),
    'taco_debug_fabric_probe': (C.c_int, [_P, _P, _P, C.c_int64, _I, _P]),
    'taco_denorm_unframe': (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'taco_griffinlim_workspace_bytes': (C.c_int64, [_I, _I]),
    'taco_griffinlim': (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'taco_fill_bernoulli': (C.c_int, [_P, C.c_int64, C.c_float, C.c_uint64, _P]),
    'taco_profile_enable': (C.c_int, [_I]),
    'taco_debug_last_cluster': (C.c_int, [_I]),
    'taco_profile_read': (C.c_int, [_I, C.POINTER(C.c_float), _I]),
    'taco_debug_profile_labels': (C.c_int, [_I, C.c_char_p, _I]),
    'taco_profile_read2': (C.c_int, [_I, C.POINTER(C.c_float), C.POINTER(C.c_double), _I]),
}
for _name, (_res, _args) in EXPORTS.items():
    _fn = getattr(_lib, _name)  # AttributeError here = the library does not export what include/taco_hip.h declares
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return _lib.taco_last_error_string().decode()


def _check(rc: int, what: str):
    if rc != 0:
        raise TacoError('%s failed (rc=%d): %s' % (what, rc, last_error()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'taco: tensor must be contiguous'
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_shape(B, Tt, Td, r, V, S=1) -> TacoShape:
    """S = num_speakers (<= 1: single-speaker model, no speaker path)."""
    return TacoShape(int(B), int(Tt), int(Td), int(r), int(V), int(S))


def version() -> int:
    return _lib.taco_version()


def param_count(shape: TacoShape) -> int:
    n = _lib.taco_param_count(C.byref(shape))
    if n < 0:
        raise TacoError('taco_param_count: ' + last_error())
    return n


def _table(fn, *args):
    n = fn(*args, None, 0)
    if n < 0:
        raise TacoError(last_error())
    rows = (TacoTensorInfo * n)()
    fn(*args, rows, n)
    return [(r.name.decode(), int(r.offset), int(r.size), tuple(r.dims[i] for i in range(r.ndim))) for r in rows]


def param_table(shape: TacoShape):
    """[(name, offset_floats, size_floats, dims)] in TF variable order."""
    return _table(_lib.taco_param_table, C.byref(shape))


def workspace_bytes(shape: TacoShape, train: bool) -> int:
    n = _lib.taco_workspace_bytes(C.byref(shape), int(train))
    if n < 0:
        raise TacoError('taco_workspace_bytes: ' + last_error())
    return n


def workspace_table(shape: TacoShape, train: bool):
    return _table(_lib.taco_workspace_table, C.byref(shape), int(train))


def conv_gemm(A, W, C_out, M, N, K, taps=1, T=None, pad_l=0, act=0, bias=None, scale=None, shift=None, residual=None,
              keep=None, Cpre=None, lda=None, ldw=None, ldc=None, ldr=None):
    T = M if T is None else T
    _check(_lib.taco_conv_gemm(ptr(A), lda or K, ptr(W), ldw or N, ptr(bias), ptr(scale), ptr(shift), ptr(residual),
                               ldr or N, ptr(keep), ptr(C_out), ldc or N, ptr(Cpre), M, N, K, taps, T, pad_l, act,
                               stream_ptr()), 'taco_conv_gemm')


def debug_gemm2_window(lo, hi):
    return int(_lib.taco_debug_gemm2_window(lo, hi))


def conv_gemm_nld(A, W, C_out, M, N, K, nld, ldw, ldc, act=0, bias=None):
    _check(_lib.taco_debug_conv_gemm_nld(ptr(A), K, ptr(W), ldw, nld, ptr(bias), ptr(C_out), ldc, M, N, K, act, stream_ptr()),
           'taco_debug_conv_gemm_nld')


def conv_gemm_ksplit(A, W, C_out, M, N, K, slabs, taps=1, T=None, pad_l=0, act=0, bias=None):
    T = M if T is None else T
    _check(_lib.taco_debug_conv_gemm_ksplit(ptr(A), K, ptr(W), N, ptr(bias), ptr(C_out), N, M, N, K, taps, T, pad_l, act,
                                            ptr(slabs), slabs.numel(), stream_ptr()), 'taco_debug_conv_gemm_ksplit')


def gemm_tn(A, dY, dW, M, N, K, taps=1, T=None, pad_l=0, accumulate=False, lda=None, ldy=None, ldw=None):
    T = M if T is None else T
    _check(_lib.taco_gemm_tn(ptr(A), lda or K, ptr(dY), ldy or N, ptr(dW), ldw or N, M, N, K, taps, T, pad_l,
                             int(accumulate), stream_ptr()), 'taco_gemm_tn')


def weight_image(W=None, taps=1, K=0, N=0, ldw=None):
    """Op-level door to the pre-split weight images (include/taco_hip.h taco_debug_weight_image).  W None: clear this thread's
    table.  Otherwise builds and registers the bf16 plane image of W (taps, K, N) and returns the image tensor (keep it alive)."""
    if W is None:   # -> launches that ran the image form since the previous clear
        return int(_lib.taco_debug_weight_image(None, 0, 0, 0, 0, None, 0, None))
    ldw = ldw or N
    need = _lib.taco_debug_weight_image(ptr(W), ldw, taps, K, N, None, 0, None)
    if need < 0:
        raise TacoError('taco_debug_weight_image: %d' % need)
    img = torch.empty(need // 2, dtype=torch.int16, device=W.device)
    rc = _lib.taco_debug_weight_image(ptr(W), ldw, taps, K, N, ptr(img), need, stream_ptr())
    _check(int(rc), 'taco_debug_weight_image')
    return img


def debug_gemm_naive(A, W, C_out, M, N, K, taps=1, T=None, pad_l=0, act=0, bias=None):
    T = M if T is None else T
    _check(_lib.taco_debug_gemm_naive(ptr(A), K, ptr(W), N, ptr(bias), ptr(C_out), N, M, N, K, taps, T, pad_l, act,
                                      stream_ptr()), 'taco_debug_gemm_naive')


def bigru_fwd(x, w, xg, out, ruc, B, T):
    """w: dict with fw/bw gates/candidate kernels and biases (TF layout)."""
    _check(_lib.taco_bigru_fwd(ptr(x), ptr(w['fw/gates/kernel']), ptr(w['fw/gates/bias']),
                               ptr(w['fw/candidate/kernel']), ptr(w['fw/candidate/bias']),
                               ptr(w['bw/gates/kernel']), ptr(w['bw/gates/bias']),
                               ptr(w['bw/candidate/kernel']), ptr(w['bw/candidate/bias']),
                               ptr(xg), ptr(out), ptr(ruc), B, T, stream_ptr()), 'taco_bigru_fwd')


def forward(shape, params, text, text_length, mel, stft, masks, s2s, out, align, loss, workspace, speaker=None):
    m = masks or {}
    _check(_lib.taco_forward(C.byref(shape), ptr(params), ptr(text), ptr(text_length), ptr(speaker), ptr(mel), ptr(stft),
                             ptr(m.get('enc_keep1')), ptr(m.get('enc_keep2')), ptr(m.get('dec_keep1')),
                             ptr(m.get('dec_keep2')), ptr(m.get('sample')), ptr(s2s), ptr(out), ptr(align), ptr(loss),
                             ptr(workspace), stream_ptr()), 'taco_forward')


def backward(shape, params, text, text_length, s2s, align, masks, grads, workspace, speaker=None):
    m = masks or {}
    _check(_lib.taco_backward(C.byref(shape), ptr(params), ptr(text), ptr(text_length), ptr(speaker), ptr(s2s), ptr(align),
                              ptr(m.get('enc_keep1')), ptr(m.get('enc_keep2')), ptr(m.get('dec_keep1')),
                              ptr(m.get('dec_keep2')), ptr(m.get('sample')), ptr(grads), ptr(workspace), stream_ptr()),
           'taco_backward')


def infer(shape, params, text, text_length, s2s, out, align, workspace, speaker=None):
    _check(_lib.taco_infer(C.byref(shape), ptr(params), ptr(text), ptr(text_length), ptr(speaker), ptr(s2s), ptr(out), ptr(align),
                           ptr(workspace), stream_ptr()), 'taco_infer')


def clip_adam_step(params, grads, m, v, lr, cap, step, scratch, gnorm_out, err_words=None):
    """err_words: int32 view of the workspace's `dec.err` tensor (first two words) or None (unguarded)."""
    assert scratch.numel() >= 256, 'taco: clip_adam scratch must hold 256 floats'
    _check(_lib.taco_clip_adam_step_guarded(ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), float(lr), float(cap),
                                            int(step), ptr(scratch), ptr(gnorm_out), ptr(err_words), stream_ptr()),
           'taco_clip_adam_step_guarded')


def clear_error(shape, train, workspace):
    _check(_lib.taco_clear_error(C.byref(shape), int(train), ptr(workspace), stream_ptr()), 'taco_clear_error')


def grad_segments(shape):
    """Float offsets [b0 .. b5] of the five gradient segments (0 embedding + encoder pre_net, 1 encoder conv bank, 2 rest of the
    encoder, 3 decoder, 4 post-net); they become final in the order 4, 3, 2, 1, 0 during taco_backward."""
    b = (C.c_int64 * 8)()
    n = _lib.taco_grad_segments(C.byref(shape), b)
    if n < 1 or n > 7:
        raise TacoError('taco_grad_segments: ' + last_error())
    return [int(x) for x in b[:n + 1]]


def wait_grad_segment(seg, stream):
    """Device-side wait of `stream` (a torch.cuda.Stream) for segment `seg` of this thread's last taco_backward."""
    _check(_lib.taco_wait_grad_segment(int(seg), C.c_void_p(stream.cuda_stream)), 'taco_wait_grad_segment')


def decoder_mode(mode=-1) -> int:
    """Process-wide decoder mode (include/taco_hip.h): 0 decoder3 fast exchange, 1 decoder3 agent-scope exchange, 2 decoder.hip.
    Sets it when mode >= 0; returns the PREVIOUS mode."""
    return _lib.taco_decoder_mode(int(mode))


def clock_probe(iters=1 << 20):
    """GHz the chip sustains under a chip-wide latency-bound load (every CU: 8 waves of dependent FMAs); host synchronisation."""
    out = torch.zeros(3, dtype=torch.int64, device='cuda')
    for _ in range(2):   # (first call: clocks ramping up)
        _check(_lib.taco_debug_clock_probe(ptr(out), int(iters), stream_ptr()), 'taco_debug_clock_probe')
    torch.cuda.synchronize()
    cyc, ticks = out[0].item(), out[1].item()
    return cyc / (ticks * 10.0) if ticks > 0 else 0.0


def fabric_probe(iters=2000, scratch_mb=768):
    """Latencies of this box that the latency-bound kernels wait on (include/taco_hip.h taco_debug_fabric_probe); host
    synchronisation.  Returns ns per granule hop (one way) in the three exchange forms, ns per dependent load (plain, L2 hit / agent scope
    on cold lines), the bandwidth one workgroup streams at, and which XCDs the probing workgroups ran on."""
    out = torch.zeros(32, dtype=torch.int64, device='cuda')
    gran = torch.zeros(512, dtype=torch.int64, device='cuda')
    scratch = torch.zeros(scratch_mb << 18, dtype=torch.int32, device='cuda')
    res = None
    for _ in range(2):   # (first call: clocks ramping up, cold TLBs)
        out.zero_(); gran.zero_()
        _check(_lib.taco_debug_fabric_probe(ptr(out), ptr(gran), ptr(scratch), int(scratch.numel() * 4), int(iters), stream_ptr()),
               'taco_debug_fabric_probe')
        torch.cuda.synchronize()
        res = out.tolist()
    n = max(1, res[6])
    ok = res[7]
    xcc = res[8:32]
    hop = lambda t: t * 10.0 / n / 2.0   # noqa: E731  (a round trip is two hops)
    return {'hop_ns_same_xcd_l2_local': hop(res[0]) if ok & 1 else None,
            'hop_ns_same_xcd_agent_scope': hop(res[1]) if ok & 2 else None,
            'hop_ns_cross_xcd': hop(res[2]) if ok & 4 else None,
            'load_ns_l2_hit': res[3] * 10.0 / n, 'load_ns_agent_scope_cold': res[4] * 10.0 / n,
            'one_cu_stream_gb_s': (8 << 20) / (res[5] * 10e-9) / 1e9 if res[5] > 0 else None,
            'xcc_of_pairs': {'l2_local': [xcc[0], xcc[8]], 'agent': [xcc[1], xcc[9]], 'cross': [xcc[2], xcc[3]]},
            'scratch_mb': scratch_mb}


def debug_spin(blocks, threads, lds_bytes, usec, stream=None):
    """Communication-kernel stand-in (co-residency tests): spins `usec` microseconds on `stream` (default: current)."""
    sp = stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
    _check(_lib.taco_debug_spin(int(blocks), int(threads), int(lds_bytes), int(usec), sp), 'taco_debug_spin')


def denorm_unframe(output, stft_mean, stft_std, r, want_spec=True, want_mag_t=False):
    """(B, Td, r*C) normalised r-frame layout -> chronological de-normalised (B, F, C) [and / or exp() transposed (B, C, F)]."""
    B, Td, RC = output.shape
    Cw = RC // r
    F = (Td // 4) * 4 * r
    spec = torch.empty(B, F, Cw, device=output.device) if want_spec else None
    mag_t = torch.empty(B, Cw, F, device=output.device) if want_mag_t else None
    _check(_lib.taco_denorm_unframe(ptr(output), ptr(stft_mean), ptr(stft_std), ptr(spec), ptr(mag_t), B, Td, r, Cw,
                                    stream_ptr()), 'taco_denorm_unframe')
    if want_spec and want_mag_t:
        return spec, mag_t
    return spec if want_spec else mag_t


def griffinlim(mag_t, phase0, n_iter=50):
    """mag_t, phase0 (B, 1025, F) -> waveform (B, 300 (F - 1)); audio.griffinlim on the GPU."""
    B, Cb, F = mag_t.shape
    assert Cb == 1025 and phase0.shape == mag_t.shape
    nbytes = _lib.taco_griffinlim_workspace_bytes(B, F)
    if nbytes < 0:
        raise TacoError('taco_griffinlim_workspace_bytes: bad shape')
    work = torch.empty(nbytes // 4, device=mag_t.device)
    wave = torch.empty(B, 300 * (F - 1), device=mag_t.device)
    _check(_lib.taco_griffinlim(ptr(mag_t), ptr(phase0), ptr(wave), ptr(work), B, F, int(n_iter), stream_ptr()),
           'taco_griffinlim')
    return wave


def fill_bernoulli(out, p_one, seed):
    _check(_lib.taco_fill_bernoulli(ptr(out), out.numel(), float(p_one), int(seed) & 0xFFFFFFFFFFFFFFFF, stream_ptr()),
           'taco_fill_bernoulli')


def last_cluster(which: int) -> int:
    return _lib.taco_debug_last_cluster(int(which))


def profile_enable(mask):
    """mask: bit c = category c (0 decoder fwd, 1 decoder bwd, 2 GEMM family, 3 bi-GRU); True = both decoder kernels."""
    _check(_lib.taco_profile_enable(3 if mask is True else int(mask)), 'taco_profile_enable')


def profile_labels(which: int):
    """What each launch currently in ring `which` was (kernel family + shape); call before profile_read."""
    cap = 1 << 19
    buf = C.create_string_buffer(cap)
    n = _lib.taco_debug_profile_labels(int(which), buf, cap)
    if n < 0:
        raise TacoError('taco_debug_profile_labels: ' + last_error())
    return buf.value.decode().split('\n')[:n]


def profile_read(which: int, with_flops=False):
    """Elapsed milliseconds (and algorithmic FLOPs) of every recorded launch of category `which` since the last read."""
    cap = 4096
    buf = (C.c_float * cap)()
    fl = (C.c_double * cap)()
    n = _lib.taco_profile_read2(which, buf, fl, cap)
    if n < 0:
        raise TacoError('taco_profile_read: ' + last_error())
    n = min(n, cap)
    ms = [buf[i] for i in range(n)]
    return (ms, [fl[i] for i in range(n)]) if with_flops else ms
