"""Host-side restatements of two index maps of tacotron_amd/csrc/gemm2.hip (no GPU): the XCD-aware tile order and the shifted
float4 epilogue for row pitches that are not multiples of 4 floats.  The kernel's arithmetic is restated line by line; the GPU
tests (test_gpu_ops.py::test_conv_gemm_v2_shifted_rows, the model parity tests) check the kernel itself."""
import itertools


def xcd_lin(bid, tiles):
    """conv_gemm2_kernel: workgroup id -> linear tile when Gemm2Args::xcd_map is set (workgroup bid runs on XCD bid % 8)."""
    q, rem, x, j = tiles >> 3, tiles & 7, bid & 7, bid >> 3
    return x * q + min(x, rem) + j


def test_xcd_tile_order_is_a_bijection_with_contiguous_ranges_per_xcd():
    for tiles in list(range(1, 300)) + [720, 728, 810, 816, 1024, 4095]:
        lins = [xcd_lin(b, tiles) for b in range(tiles)]
        assert sorted(lins) == list(range(tiles))
        for x in range(8):
            mine = [xcd_lin(b, tiles) for b in range(x, tiles, 8)]      # dispatch order on XCD x
            assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True
        # the n-tiles of one m-tile are neighbours in the list, hence (except across a range boundary) on one XCD
        starts = sorted(xcd_lin(x, tiles) for x in range(min(8, tiles)))
        assert starts[0] == 0


def shifted_stores(m, n0, N, ldc):
    """Stores of one output row of one 128-column tile (two half-waves of 32 lanes own the same columns for different rows):
    returns (list of (column, width) stores).  Mirrors the `P.flags & 8` epilogue."""
    ldm = ldc & 3
    sft = (4 - (((m & 3) * ldm) & 3)) & 3
    out = []
    for li in range(32):
        n = n0 + 4 * li
        if n >= N:
            continue                                    # `if (n >= P.N) return;` in front of the epilogue
        c0 = n + sft
        if li < 31 and c0 + 3 < N:
            out.append((c0, 4))
        else:
            out += [(n + j, 1) for j in range(4) if j >= sft and n + j < N]
            if li < 31:
                out += [(n + 4 + j, 1) for j in range(3) if j < sft and n + 4 + j < N]
        if li == 0:
            out += [(n + j, 1) for j in range(3) if j < sft and n + j < N]
    return out


def test_shifted_epilogue_covers_every_column_once_with_aligned_vectors():
    for ldc, N, m in itertools.product([1025, 1026, 1027, 133, 518, 1030], [1025, 1024, 131, 514, 5, 128, 129], range(8)):
        if N > ldc:
            continue
        written = []
        for n0 in range(0, N, 128):
            for col, width in shifted_stores(m, n0, N, ldc):
                if width == 4:
                    assert (m * ldc + col) % 4 == 0, 'float4 store not 16-byte aligned'
                written += list(range(col, col + width))
        assert sorted(written) == list(range(N)), (ldc, N, m)


def bank_tap(g):
    """conv_gemm2_kernel, ConvGemmProblem::bank_filters: global tap index -> (width f, tap j of it, row shift, column block of d bank)."""
    f, j = 1, g
    while j >= f:
        j -= f
        f += 1
    return f, j, j - ((f - 1) - (f - 1) // 2), (f - 1) * 128


def test_bank_gather_tap_walk_matches_the_K_separate_problems():
    """The K-problem form this replaces: width f has f taps, backward pad (f-1) - (f-1)//2, reads column block f-1, and its
    transposed kernel starts f(f-1)/2 taps into the contiguous block.  Also the incremental walk the kernel does at tap changes
    and the chunking of the (tap, k-tile) sequence used by cbhg_bwd."""
    for F in (8, 16):
        expect = [(f, j, j - ((f - 1) - (f - 1) // 2), (f - 1) * 128) for f in range(1, F + 1) for j in range(f)]
        assert len(expect) == F * (F + 1) // 2
        assert [bank_tap(g) for g in range(len(expect))] == expect
        f, j = 1, 0                      # the walk of `advance()`
        for g in range(len(expect)):
            assert (f, j) == expect[g][:2]
            j += 1
            if j == f:
                f, j = f + 1, 0
        pad_max = (F - 1) - (F - 1) // 2
        assert all(-pad_max <= e[2] <= pad_max for e in expect)      # pad_l of the launch covers every shift
        nit = len(expect) * 4            # 32-deep k-tiles of K = 128
        for mtiles in (1, 50, 90):
            S = max(1, min(512 // mtiles, 16))
            per = -(-nit // S)
            chunks = [(c * per, min(nit, (c + 1) * per)) for c in range(-(-nit // per))]
            assert chunks[0][0] == 0 and chunks[-1][1] == nit and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
            assert all(hi > lo for lo, hi in chunks) and len(chunks) <= 16
