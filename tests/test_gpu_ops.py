"""GPU: op-level parity of the HIP kernels (through the C ABI) against NumPy fp64 restatements.
Tolerances (fp32 kernel vs fp64 oracle): rel-L2 <= 2e-6 * sqrt(K)-ish, stated per test."""
import numpy as np
import pytest
import torch

from oracle import taco_numpy as on
from tests.util import report

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to('cuda', dtype).contiguous()


def conv_ref(A, W, bias, T, pad_l, act, keep=None, scale=None, shift=None, residual=None):
    """A (M,K) with M = B*T; W (taps,K,N)."""
    M, K = A.shape
    taps, _, N = W.shape
    B = M // T
    x = A.reshape(B, T, K)
    pad_r = taps - 1 - pad_l
    xp = np.pad(x, ((0, 0), (max(pad_l, 0), max(pad_r, 0)), (0, 0)))
    y = np.zeros((B, T, N))
    for j in range(taps):
        sh = j - pad_l
        for t in range(T):
            st = t + sh
            if 0 <= st < T:
                y[:, t] += x[:, st] @ W[j]
    y = y.reshape(M, N)
    if bias is not None:
        y = y + bias
    if act == 1:
        y = np.maximum(y, 0)
    elif act == 2:
        y = on.sigmoid(y)
    elif act == 3:
        y = np.tanh(y)
    if keep is not None:
        y = y * keep * 2
    pre = y.copy()
    if scale is not None:
        y = y * scale + (shift if shift is not None else 0)
    if residual is not None:
        y = y + residual
    return y, pre


def test_naive_gemm_plumbing(built_lib):
    rng = np.random.default_rng(0)
    M, N, K = 37, 24, 19
    A, W, b = rng.standard_normal((M, K)), rng.standard_normal((1, K, N)), rng.standard_normal(N)
    C = torch.zeros(M, N, device='cuda')
    built_lib.debug_gemm_naive(dev(A), dev(W), C, M, N, K, bias=dev(b), act=1)
    ref, _ = conv_ref(A, W, b, M, 0, 1)
    r, m = report('naive gemm', C.cpu().numpy(), ref)
    assert r < 1e-6


CASES = [
    # M, T, N, K, taps, pad_l, act, extras
    (128, 128, 128, 128, 1, 0, 0, ''),
    (256, 64, 128, 64, 3, 1, 1, 'bias'),
    (70, 35, 80, 20, 3, 1, 1, 'bias,scale,residual,pre'),
    (90, 9, 128, 128, 16, 7, 1, 'bias'),
    (90, 9, 128, 80, 8, 3, 1, 'bias'),
    (64, 64, 1025, 256, 1, 0, 0, 'bias'),
    (200, 200, 256, 256, 1, 0, 1, 'bias,keep'),
    (6400, 200, 256, 256, 1, 0, 2, 'bias'),        # big-tile path
    (3000, 3000, 384, 136, 1, 0, 3, 'bias'),       # big-tile path with ragged edges
    (36, 9, 128, 128, 5, 2, 0, 'residual'),
    (36, 9, 80, 128, 6, 3, 0, ''),                 # backward-style pad (k-1 - (k-1)//2)
]


@pytest.mark.parametrize('case', CASES, ids=[str(c[:6]) for c in CASES])
def test_conv_gemm(built_lib, case):
    M, T, N, K, taps, pad_l, act, extras = case
    rng = np.random.default_rng(hash(case[:6]) % 2**31)
    A = rng.standard_normal((M, K))
    W = rng.standard_normal((taps, K, N)) / np.sqrt(K * taps)
    bias = rng.standard_normal(N) if 'bias' in extras else None
    keep = rng.integers(0, 2, (M, N)).astype(np.uint8) if 'keep' in extras else None
    scale = rng.standard_normal(N) if 'scale' in extras else None
    shift = rng.standard_normal(N) if 'scale' in extras else None
    res = rng.standard_normal((M, N)) if 'residual' in extras else None
    C = torch.full((M, N), float('nan'), device='cuda')
    Cpre = torch.full((M, N), float('nan'), device='cuda') if 'pre' in extras else None
    built_lib.conv_gemm(dev(A), dev(W), C, M, N, K, taps=taps, T=T, pad_l=pad_l, act=act,
                        bias=None if bias is None else dev(bias), scale=None if scale is None else dev(scale),
                        shift=None if shift is None else dev(shift), residual=None if res is None else dev(res),
                        keep=None if keep is None else dev(keep, torch.uint8), Cpre=Cpre)
    ref, pre = conv_ref(A, W, bias, T, pad_l, act, keep, scale, shift, res)
    r, m = report('conv_gemm %s' % (case[:7],), C.cpu().numpy(), ref)
    assert r < 5e-6
    if Cpre is not None:
        assert report('  Cpre', Cpre.cpu().numpy(), pre)[0] < 5e-6


V2_CASES = [c for c in CASES if c[2] % 4 == 0 and c[3] % 4 == 0] + [
    (1000, 200, 300, 132, 3, 1, 1, 'bias,keep,scale,residual,pre'),   # every ragged edge at once: M, N, K tails, 5 sequences
    (640, 40, 128, 128, 16, 7, 1, 'bias'),                            # widest conv-bank problem
    (720, 360, 1024, 256, 3, 1, 0, ''),                               # post-net dpool-like: many n-tiles
    (520, 520, 260, 2048, 1, 0, 0, 'bias'),                           # deep K (64 k-tiles of 32)
]


# (MFMA form, k-tile depth x ring stages): 'f32' = v_mfma_f32_32x32x2_f32 (rounds 2-4), 'bx' = the bf16x3 form of round 5
V2_VARIANTS = [('0', '32x2'), ('0', '32x3'), ('0', '16x3'), ('0', '16x4'), ('0', '16x5'), ('1', '16x3'), ('1', '16x4'), ('1', '16x5')]
V2_IDS = ['%s-%s' % ('bx' if b == '1' else 'f32', v) for b, v in V2_VARIANTS]


@pytest.mark.parametrize('variant', V2_VARIANTS, ids=V2_IDS)
@pytest.mark.parametrize('case', V2_CASES, ids=[str(c[:6]) for c in V2_CASES])
def test_conv_gemm_v2(built_lib, case, variant, monkeypatch):
    """gemm2.hip (DMA-staged, swizzled kernel) forced for every shape that meets its contract, in all its instantiations -- the
    fp32 MFMA form in five (k-tile depth x ring stages) shapes and the bf16x3 form (fp32 operands split into three bf16 planes in
    registers, six v_mfma_f32_32x32x16_bf16 per sub-tile, TACO_GEMM2_BF16X) in three: K / N / M tails as out-of-range buffer
    offsets (zeros), tap shifts across sequence boundaries, the float4 and the scalar epilogue.  Same tolerance for both forms."""
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_BF16X', variant[0])
    monkeypatch.setenv('TACO_GEMM2_VARIANT', variant[1])
    test_conv_gemm(built_lib, case)


WIMG_CASES = [c for c in V2_CASES if c[7] in ('', 'bias')] + [
    (1000, 200, 300, 132, 3, 1, 1, 'bias'),      # ragged: N and K tails are zeros of the image
    (640, 40, 80, 128, 8, 4, 0, ''),             # N = 80 (post-net bank gather's width)
    (384, 384, 1028, 256, 1, 0, 0, 'bias'),      # nine n-tiles, the last one 4 columns wide
]


@pytest.mark.parametrize('ns', ['3', '4'])
@pytest.mark.parametrize('case', WIMG_CASES, ids=[str(c[:6]) for c in WIMG_CASES])
def test_conv_gemm_weight_image(built_lib, case, ns, monkeypatch):
    """Round 6: the bf16x3 form with the weight operand read from a pre-split plane image (csrc/kernels.h "weight images": planes
    formed once per weight tensor by weight_image_kernel, DMA'd as 12 KB fragment-ordered tiles, no B split in the GEMM).  The
    image holds exactly the planes the in-register split forms and the MFMAs run in the same order, so the result must be BIT
    IDENTICAL to the in-register bf16x3 form (TACO_GEMM2_BSPLIT=0) -- and within the suite's 5e-6 of fp64.  Three- and four-stage
    rings; K / N tails, taps, several n-tiles."""
    M, T, N, K, taps, pad_l, act, extras = case
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_BF16X', '1')
    monkeypatch.setenv('TACO_GEMM2_BI_NS', ns)
    rng = np.random.default_rng(hash(case[:6]) % 2**31)
    A = rng.standard_normal((M, K))
    W = rng.standard_normal((taps, K, N)) / np.sqrt(K * taps)
    bias = rng.standard_normal(N) if 'bias' in extras else None
    dA, dW, db = dev(A), dev(W), (None if bias is None else dev(bias))
    ref, _ = conv_ref(A, W, bias, T, pad_l, act)
    built_lib.weight_image(None)
    C0 = torch.full((M, N), float('nan'), device='cuda')
    built_lib.conv_gemm(dA, dW, C0, M, N, K, taps=taps, T=T, pad_l=pad_l, act=act, bias=db)      # no image registered: in-register split
    assert built_lib.weight_image(None) == 0
    img = built_lib.weight_image(dW, taps=taps, K=K, N=N)
    C1 = torch.full((M, N), float('nan'), device='cuda')
    built_lib.conv_gemm(dA, dW, C1, M, N, K, taps=taps, T=T, pad_l=pad_l, act=act, bias=db)
    torch.cuda.synchronize()
    assert built_lib.weight_image(None) == 1, 'the launch did not take the image form'
    del img
    assert report('conv_gemm image form %s' % (case[:7],), C1.cpu().numpy(), ref)[0] < 5e-6
    assert torch.equal(C0, C1), 'image form differs from the in-register split: max |d| = %g' % float((C0 - C1).abs().max())


@pytest.mark.parametrize('scale', [1.0, 1e-12, 3e7], ids=['unit', 'tiny', 'huge'])
def test_bf16x3_products_are_fp32_grade(built_lib, scale, monkeypatch):
    """The bf16x3 form of gemm2.hip against the fp32 MFMA form of the same kernel and an fp64 product: a deep reduction (K = 2048
    x 3 taps) over heavy-tailed operands (log-normal magnitudes over ~5 decades; gradient-like scales included).  The split is
    exact (x = h + m + l) and the six retained plane products cover everything above 2^-24 of a product; what the bf16
    instruction adds is its 16-term internal sum (one alignment per 16 products instead of per 2).  Measured on MI355X: rel-L2
    1.8e-7 vs 1.4e-7 of the fp32 form on these operands.  Asserted: <= 2 x the fp32 form's error, and both <= the suite's 5e-6
    (i.e. 25 x below the tolerance every GEMM test of this file states)."""
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_VARIANT', '16x4')
    rng = np.random.default_rng(11)
    M, T, N, K, taps = 512, 128, 256, 2048, 3
    A = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K)) * 3)).astype(np.float32) * np.float32(scale)
    W = (rng.standard_normal((taps, K, N)) * np.exp(rng.standard_normal((taps, K, N)) * 3) / np.sqrt(K * taps)).astype(np.float32)
    ref, _ = conv_ref(A.astype(np.float64), W.astype(np.float64), None, T, 1, 0)
    err = {}
    for bx in ('0', '1'):
        monkeypatch.setenv('TACO_GEMM2_BF16X', bx)
        C = torch.full((M, N), float('nan'), device='cuda')
        before = built_lib.debug_gemm2_window(0, 1 << 30)
        built_lib.conv_gemm(dev(A), dev(W), C, M, N, K, taps=taps, T=T, pad_l=1, act=0)
        assert built_lib.debug_gemm2_window(0, 1 << 30) == 1, 'the launch did not go to gemm2.hip'
        err[bx] = report('gemm2 %s scale=%g' % ('bf16x3' if bx == '1' else 'fp32  ', scale), C.cpu().numpy(), ref)
    assert err['0'][0] < 5e-6 and err['1'][0] < 5e-6
    assert err['1'][0] <= 2.0 * err['0'][0] and err['1'][1] <= 2.5 * err['0'][1]


def _ones_mantissa(rng, shape, spread=2):
    """Same-signed fp32 values with ALL mantissa bits set, (2 - 2^-23) * 2^e with e ~ U{-spread..spread}: each of the three bf16
    planes of the split is as large as it can be relative to the one above, and with one sign the three dropped plane products
    (m l, l m, l l) of the bf16x3 form cannot cancel -- its coherent worst case."""
    e = rng.integers(-spread, spread + 1, shape)
    return (np.float32(2.0 - 2.0 ** -23) * np.exp2(e).astype(np.float32)).astype(np.float32)


def _adversarial_case(rng, kernel, kind, depth):
    """(run, ref): the NN kernel of gemm2.hip (M x depth @ depth x N) or the weight-gradient kernel of gemm.hip (dW = A^T dY over
    `depth` rows) on same-signed all-ones-mantissa or heavy-tailed operands."""
    def operand(shape, norm):
        if kind == 'heavy-tailed':
            return (rng.standard_normal(shape) * np.exp(rng.standard_normal(shape) * 3) / norm).astype(np.float32)
        return _ones_mantissa(rng, shape)
    if kernel == 'nn':
        M, N, K = 512, 256, depth
        A, W = operand((M, K), 1.0), operand((1, K, N), np.sqrt(K))
        ref = A.astype(np.float64) @ W[0].astype(np.float64)

        def run(built_lib):
            C = torch.full((M, N), float('nan'), device='cuda')
            built_lib.debug_gemm2_window(0, 1 << 30)
            built_lib.conv_gemm(dev(A), dev(W), C, M, N, K, taps=1, T=M, pad_l=0, act=0)
            assert built_lib.debug_gemm2_window(0, 1 << 30) == 1, 'the launch did not go to gemm2.hip'
            return C.cpu().numpy()
    else:
        M, N, K = depth, 256, 512          # dW (K, N) = A^T (K, M) dY (M, N): the reduction runs over the M = depth rows
        A, dY = operand((M, K), 1.0), operand((M, N), np.sqrt(M))
        ref = A.astype(np.float64).T @ dY.astype(np.float64)

        def run(built_lib):
            dW = torch.full((1, K, N), float('nan'), device='cuda')
            built_lib.gemm_tn(dev(A), dev(dY), dW, M, N, K, taps=1, T=M, pad_l=0, accumulate=False)
            return dW[0].cpu().numpy()
    return run, ref


@pytest.mark.parametrize('kind', ['ones-mantissa-same-sign', 'heavy-tailed'])
@pytest.mark.parametrize('kernel,depth', [('nn', 6144), ('nn', 2048), ('tn', 6144), ('tn-deterministic', 6144)],
                         ids=['nn-6144', 'nn-2048-deepest-bf16x3-chain', 'tn-6144', 'tn-6144-one-workgroup-per-tile'])
def test_bf16x3_adversarial_operands(built_lib, kernel, depth, kind, monkeypatch):
    """Worst-case rather than statistical evidence for the products of the big GEMM kernels (VERDICT r5 #9 / next #7a), on BOTH
    kernels that have a bf16x3 form: the NN kernel of gemm2.hip and the weight-gradient kernel of gemm.hip (gemm_tn).
    `ones-mantissa-same-sign`: every operand positive with all 23 mantissa bits set -- no cancellation anywhere: the three dropped
    plane products (m l, l m, l l) are one-signed, and the accumulator grows linearly, which is what exposes how the matrix pipe adds
    (it rounds every product by itself onto the accumulator's grid: tools/micro/mfma_bf16_probe.hip); `heavy-tailed`: log-normal
    magnitudes over ~5 decades.  What the library does about it (bf16x3.h): the low-order plane products have an accumulator of
    their own (`mfma6_2`), and the bf16x3 form is used for accumulation chains of at most 2048 products (`bf16x_max_chain`); deeper
    ones run the fp32 MFMA instruction.  Asserted, default settings against an fp64 product: **rel-L2 <= 4e-7** (the verdict's bar)
    in every case -- depth 6144 (encoder proj1's depth: NN routed to the fp32 instruction, gemm_tn's split row ranges of 384 rows
    on bf16x3, its single-workgroup form routed) and depth 2048, the deepest chain the bf16x3 form is used for (measured 3.2e-7 on
    same-signed full mantissas, 2.0e-7 heavy-tailed) -- and <= 3 x the error of the forced fp32 MFMA form with a floor of 1.2e-7
    (on same-signed full mantissas the fp32 instruction itself measures 2e-8, a level the split form does not reach: the ratio is
    printed; heavy-tailed NN at depth 6144: the fp32 instruction measures 4.3e-7 and is what runs)."""
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_VARIANT', '16x4')
    if kernel == 'tn-deterministic':
        monkeypatch.setenv('TACO_DETERMINISTIC', '1')
    run, ref = _adversarial_case(np.random.default_rng(23), kernel[:2], kind, depth)
    err = {}
    for bx in ('0', '1'):
        monkeypatch.setenv('TACO_GEMM2_BF16X', bx)
        err[bx] = report('%s depth %d %s %s' % (kernel, depth, kind, 'default' if bx == '1' else 'fp32  '), run(built_lib), ref)
    print('  %s depth %d / %s: rel-L2 default %.2e vs forced fp32 MFMA %.2e (ratio %.2f); 4e-7 bar: default %s, fp32 MFMA %s'
          % (kernel, depth, kind, err['1'][0], err['0'][0], err['1'][0] / max(err['0'][0], 1e-30),
             'met' if err['1'][0] <= 4e-7 else 'NOT met', 'met' if err['0'][0] <= 4e-7 else 'NOT met'))
    assert err['0'][0] < 5e-6 and err['1'][0] < 5e-6
    assert err['1'][0] <= max(4e-7, 1.05 * err['0'][0])       # the 4e-7 bar (or the fp32 instruction's own error where that is what runs)
    assert err['1'][0] <= 3.0 * max(err['0'][0], 1.2e-7)


def test_bf16x3_chain_bound_is_what_keeps_deep_same_signed_sums_fp32_grade(built_lib, monkeypatch):
    """The reason for `bf16x_max_chain` (bf16x3.h), kept executable: with the bound lifted, a 6144-deep sum of same-signed
    full-mantissa products on the bf16x3 form measures 2.4e-6 (two accumulators; 1.7e-5 with one) against 5e-8 of the fp32
    instruction -- outside the 4e-7 bar the bounded path meets -- while mixed-sign operands are MORE accurate than the fp32
    instruction at any depth (0.3-0.4 x; tools/bf16x3_chain_probe.py, profiles/r06_bf16x3_acc2.txt).  If this test ever FAILS
    because the forced form has become accurate, the bound can go."""
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_VARIANT', '16x4')
    monkeypatch.setenv('TACO_GEMM2_BF16X', '1')
    run, ref = _adversarial_case(np.random.default_rng(23), 'nn', 'ones-mantissa-same-sign', 6144)
    bounded = report('nn 6144 same-signed, default bound', run(built_lib), ref)[0]
    monkeypatch.setenv('TACO_BF16X_MAX_CHAIN', str(1 << 30))
    forced = report('nn 6144 same-signed, bf16x3 forced ', run(built_lib), ref)[0]
    assert bounded <= 4e-7 and forced > 10 * bounded and forced > 4e-7


@pytest.mark.parametrize('case', [(1000, 200, 128, 2048, 3, 1, 1), (520, 130, 256, 1024, 3, 1, 0), (300, 300, 132, 516, 1, 0, 3)],
                         ids=['enc-proj1-like', 'post-proj1-like', 'ragged'])
def test_conv_gemm_ksplit(built_lib, case):
    """Tall-skinny projections: (tap, k) chunks as independent workgroups of the gemm2 kernel + ordered slab sum; twice, to
    pin run-to-run bit reproducibility (no atomics anywhere on this path)."""
    M, T, N, K, taps, pad_l, act = case
    rng = np.random.default_rng(7)
    A = rng.standard_normal((M, K))
    W = rng.standard_normal((taps, K, N)) / np.sqrt(K * taps)
    bias = rng.standard_normal(N)
    slabs = torch.empty(4 * M * N, device='cuda')
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float('nan'), device='cuda')
        built_lib.conv_gemm_ksplit(dev(A), dev(W), C, M, N, K, slabs, taps=taps, T=T, pad_l=pad_l, act=act, bias=dev(bias))
        outs.append(C.clone())
    ref, _ = conv_ref(A, W, bias, T, pad_l, act)
    assert report('conv_gemm_ksplit %s' % (case,), outs[0].cpu().numpy(), ref)[0] < 5e-6
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('variant', ['16x4', '32x2', 'bx'])
@pytest.mark.parametrize('N,ldc,act', [(1025, 1025, 0), (1025, 1027, 1), (514, 518, 0), (131, 133, 3), (1024, 1025, 0), (1025, 1030, 0), (133, 135, 0),
                                       (262, 263, 0)],
                         ids=['dense-1025', 'pitch-3mod4', 'pitch-2mod4', 'narrow', 'full-tiles', 'pitch-2mod4-odd-N', 'tail-5-of-a-tile',
                              'tail-6-of-a-tile'])
def test_conv_gemm_v2_shifted_rows(built_lib, N, ldc, act, variant, monkeypatch):
    """gemm2.hip's shifted float4 epilogue: output rows whose pitch is not a multiple of 4 floats (the final dense layer writes
    (B*F, 1025) from a weight copy padded to 1028 columns, tacotron.py:148 / model.hip).  Every pitch residue, the N tail inside
    a tile, an M tail; nothing may land outside the N columns of a row (sentinel in the pitch padding and behind the last row),
    and the kernel must really be the DMA one (the old kernel is switched off for the call)."""
    monkeypatch.setenv('TACO_GEMM2_MIN_TILES', '1')
    monkeypatch.setenv('TACO_GEMM2_BF16X', '1' if variant == 'bx' else '0')
    monkeypatch.setenv('TACO_GEMM2_VARIANT', '16x4' if variant == 'bx' else variant)
    rng = np.random.default_rng(N * 7 + ldc)
    M, K = 333, 96
    nld = (N + 3) // 4 * 4
    A = rng.standard_normal((M, K))
    W = rng.standard_normal((1, K, N)) / np.sqrt(K)
    Wp = np.zeros((K, nld)); Wp[:, :N] = W[0]
    bias = rng.standard_normal(N)
    flat = torch.full((M * ldc + 64,), 7.5, device='cuda')
    before = built_lib.debug_gemm2_window(0, 1 << 30)
    built_lib.conv_gemm_nld(dev(A), dev(Wp), flat, M, N, K, nld, nld, ldc, act=act, bias=dev(bias))
    assert built_lib.debug_gemm2_window(0, 1 << 30) == 1, 'the launch did not go to gemm2.hip'
    ref, _ = conv_ref(A, W, bias, M, 0, act)
    C = flat[:M * ldc].view(M, ldc)
    assert report('shifted rows N=%d ldc=%d %s' % (N, ldc, variant), C[:, :N].cpu().numpy(), ref)[0] < 5e-6
    assert bool((C[:, N:] == 7.5).all()) and bool((flat[M * ldc:] == 7.5).all())


def test_conv_gemm_strided_unaligned(built_lib):
    """lda not a multiple of 4 (1025-wide rows) exercises the scalar-load path; ldc > N exercises column offsets."""
    rng = np.random.default_rng(5)
    M, K, N = 96, 1025, 256
    A = rng.standard_normal((M, K))
    W = rng.standard_normal((1, K, N)) / 32
    flat = torch.zeros(M * 300 + 20, device='cuda')          # C = flat[20:] viewed with ldc = 300
    built_lib.conv_gemm(dev(A), dev(W), flat[20:], M, N, K, lda=1025, ldc=300)
    C = flat[:M * 300].view(M, 300)
    ref, _ = conv_ref(A, W, None, M, 0, 0)
    assert report('unaligned lda', C[:, 20:276].cpu().numpy(), ref)[0] < 5e-6
    assert float(C[:, :20].abs().max()) == 0 and float(C[:, 276:].abs().max()) == 0


TN_CASES = [(128, 128, 128, 128, 1, 0), (360, 36, 80, 128, 3, 1), (5760, 180, 256, 512, 1, 1), (400, 40, 128, 128, 1, -1),
            (90, 9, 128, 128, 16, 7), (512, 512, 1025, 256, 1, 0), (77, 77, 20, 36, 1, 0),
            # second-generation kernel (gemm2.hip gemm_tn2: M >= 512, K, N >= 96, vector contract)
            (1600, 200, 128, 128, 16, 7), (1200, 40, 132, 260, 3, 1), (1440, 360, 256, 1024, 3, 1), (2000, 2000, 100, 96, 1, 0),
            (640, 20, 128, 128, 8, 4),
            # K no multiple of the 64-row tile: the taps are merged into K (GemmTnArgs::ktap); post-net conv bank shapes
            (720, 360, 128, 80, 8, 3), (800, 200, 128, 80, 5, 2), (360, 90, 80, 36, 2, 0)]


@pytest.mark.parametrize('tn2', ['0', '1'], ids=['gemm_tn', 'opt-in-tn2'])
@pytest.mark.parametrize('case', TN_CASES, ids=[str(c) for c in TN_CASES])
def test_gemm_tn(built_lib, case, tn2, monkeypatch):
    """Weight-gradient GEMM; with TACO_TN2=1 the eligible shapes run on the opt-in second-generation kernel (gemm2.hip)."""
    monkeypatch.setenv('TACO_TN2', tn2)
    M, T, N, K, taps, pad_l = case
    rng = np.random.default_rng(hash(case) % 2**31)
    A = rng.standard_normal((M, K))
    dY = rng.standard_normal((M, N))
    B = M // T
    ref = np.zeros((taps, K, N))
    x = A.reshape(B, T, K)
    y = dY.reshape(B, T, N)
    for j in range(taps):
        sh = j - pad_l
        for t in range(T):
            st = t + sh
            if 0 <= st < T:
                ref[j] += x[:, st].T @ y[:, t]
    dW = torch.full((taps, K, N), 7.0, device='cuda')
    built_lib.gemm_tn(dev(A), dev(dY), dW, M, N, K, taps=taps, T=T, pad_l=pad_l, accumulate=False)
    assert report('gemm_tn %s' % (case,), dW.cpu().numpy(), ref)[0] < 5e-6
    built_lib.gemm_tn(dev(A), dev(dY), dW, M, N, K, taps=taps, T=T, pad_l=pad_l, accumulate=True)
    assert report('  accumulate', dW.cpu().numpy(), 2 * ref)[0] < 5e-6


@pytest.mark.parametrize('B,T', [(2, 9), (3, 50), (32, 200)])
def test_bigru_fwd(built_lib, B, T):
    rng = np.random.default_rng(B * 100 + T)
    p = {}
    for d in ('fw', 'bw'):
        p['g/%s/gates/kernel' % d] = rng.uniform(-0.15, 0.15, (256, 256))
        p['g/%s/gates/bias' % d] = 1 + rng.uniform(-0.3, 0.3, 256)
        p['g/%s/candidate/kernel' % d] = rng.uniform(-0.15, 0.15, (256, 128))
        p['g/%s/candidate/bias' % d] = rng.uniform(-0.3, 0.3, 128)
    x = rng.standard_normal((B, T, 128))
    ref = on.bigru(x, p, 'g/')
    w = {k[2:]: dev(v) for k, v in p.items()}
    xg = torch.zeros(B, T, 768, device='cuda')
    out = torch.zeros(B, T, 256, device='cuda')
    ruc = torch.zeros(B, T, 768, device='cuda')
    built_lib.bigru_fwd(dev(x), w, xg, out, ruc, B, T)
    r, m = report('bigru_fwd B=%d T=%d' % (B, T), out.cpu().numpy(), ref)
    assert r < 2e-5 and m < 1e-4


def test_bernoulli(built_lib):
    out = torch.zeros(1 << 20, dtype=torch.uint8, device='cuda')
    built_lib.fill_bernoulli(out, 0.5, 123)
    a = out.cpu().numpy().copy()
    assert set(np.unique(a)) == {0, 1} and abs(a.mean() - 0.5) < 3e-3
    built_lib.fill_bernoulli(out, 0.5, 123)
    assert np.array_equal(a, out.cpu().numpy())
    built_lib.fill_bernoulli(out, 0.5, 124)
    b = out.cpu().numpy()
    assert abs((a == b).mean() - 0.5) < 3e-3           # independent streams
    built_lib.fill_bernoulli(out, 0.25, 9)
    assert abs(out.float().mean().item() - 0.25) < 3e-3
    odd = torch.zeros(1001, dtype=torch.uint8, device='cuda')
    built_lib.fill_bernoulli(odd, 1.0, 1)
    assert int(odd.sum()) == 1001


@pytest.mark.parametrize('gscale', [0.01, 30.0], ids=['noclip', 'clip'])
def test_clip_adam_step(built_lib, gscale):
    rng = np.random.default_rng(1)
    n = 100003
    p0 = rng.standard_normal(n)
    p = {'w': p0.copy()}
    m = {'w': np.zeros(n)}
    v = {'w': np.zeros(n)}
    P, Mm, Vv = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    scratch, gn_out = torch.zeros(256, device='cuda'), torch.zeros(1, device='cuda')
    for step in (1, 2, 3):
        g = rng.standard_normal(n) * gscale
        gn = on.clip_adam_step(p, {'w': g}, m, v, step, 5e-4)
        built_lib.clip_adam_step(P, dev(g), Mm, Vv, 5e-4, 5.0, step, scratch, gn_out)
        assert abs(gn_out.item() - gn) < 1e-4 * gn
    assert report('adam params', P.cpu().numpy(), p['w'])[1] < 2e-6
    assert report('adam m', Mm.cpu().numpy(), m['w'])[0] < 1e-5
    # fp32 (1 - 0.999f) carries a 4.7e-5 relative rounding error, exactly as TF's fp32 ApplyAdam kernel does
    assert report('adam v', Vv.cpu().numpy(), v['w'])[0] < 1e-4
