// gemm2.hip -- second-generation NN conv-GEMM for the big feed-forward launches (same contract as conv_gemm_kernel in gemm.hip:
//   C = post(act(sum_tap shift_tap(A) . W[tap] + bias)), tf.layers.dense / conv1d 'same', ops.py:54-60,80-86, tacotron.py:40-43,148).
//
// What changed against gemm.hip (47 % matrix-pipe utilisation, one barrier per 16-deep k-tile, register-staged
// loads, transposing ds_write_b32 stores):
//   * both operands travel HBM/L2 -> LDS by direct DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction): no staging
//     registers, no ds_write pass.  Round 4: through a buffer descriptor per operand -- per-lane byte offsets are set up once
//     (per tap for the shifted rows), the k-tile / tap advance is the instruction's SGPR offset, and masked elements (rows
//     outside the sequence for a shifted tap, the K / N / M tails) carry an out-of-range offset, for which the hardware
//     deposits zeros: a piece costs ~2 VALU + 3 SALU instead of ~14 VALU + a readfirstlane (the DMA issue was 13 % of a
//     4096^3 launch, profiles/r04_gemm_lab.txt).
//   * k-tiles are 32 deep (one barrier per 64 MFMAs per wave instead of per 32) in an NS-stage LDS ring with a COUNTED
//     s_waitcnt vmcnt (the DMA of the next tile(s) stays in flight across the raw s_barrier).
//   * A tile image = the global rows themselves (128 B per row, full cache lines); the 16-byte slots of a row are XOR-swizzled
//     on the SOURCE side (lane -> which slot it fetches) and on the READ side, so the fragment ds_read_b128 is conflict free
//     (MI355X guide, LDS table: ds_read_b128 is served in 4 groups of 16 lanes).
//   * one ds_read_b128 feeds FOUR MFMAs: lane (i, kh) holds A[i][4 consecutive k]; MFMA c multiplies k = 8p + 4kh + c, and
//     the B fragment is read as B[k][4 li .. 4 li + 3] = the lane's column of FOUR interleaved 32-column sub-tiles (n = 4 li + j).
//     A wave owns 32 rows x 128 columns; per 32-deep k-tile: 4 + 16 ds_read_b128, 64 v_mfma_f32_32x32x2_f32.
//   * the interleaved column map makes the epilogue a float4 store per (row, lane): 512 contiguous bytes per row.
//   * grouped launch: every (problem, m-tile, n-tile) of a batch is one workgroup of ONE grid, longest problems first.
#include <algorithm>
#include <type_traits>

#include "bf16x3.h"
#include "common.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) float g_zero4[4];   // zero-initialised device word: source of every masked DMA lane

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same DMA through a buffer descriptor: the address is  base + voffset (per lane) + soffset (SGPR), and a lane whose
// voffset + soffset reaches num_records deposits ZEROS (tools/micro/bufload_lds.hip checks both on gfx950).  Everything that
// changes from k-tile to k-tile goes into soffset, so a piece costs a scalar add instead of a 64-bit per-lane address.
constexpr int kOOB = (int)0x80000000;   // >= num_records (0x7fffffff) whatever soffset is
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, int voffset, int soffset, float* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// LDS fragment read the compiler's waitcnt pass does not see (it would wait lgkmcnt(0) right in front of the first use, i.e.
// expose the whole LDS latency every few MFMAs): the kernel counts its own lgkmcnt.  There is NO scalar-memory load inside
// the k-loop (everything is hoisted into registers first), so lgkmcnt counts exactly these reads, in issue order.
template <int OFF>
__device__ __forceinline__ f32x4 dsr128(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint32_t lds_off(const float* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// Issue order of the fragment reads of one k-tile (S = 4 NP steps, step s = (p, c) = (s / 4, s % 4)):
//   prologue  a(0) b(0) b(1);   after the MFMAs of step s:  b(s + 2)  [s + 2 < S],  then a(s / 4 + 1)  [s % 4 == 1, next p exists]
// wait_count(s) = reads issued before step s's MFMAs that are YOUNGER than the youngest read step s needs.
constexpr int rd_n(int u, int S, int NP) { return (u + 2 < S ? 1 : 0) + ((u % 4 == 1 && u / 4 + 1 < NP) ? 1 : 0); }
constexpr int rd_before(int s, int S, int NP) {
  int n = 3;
  for (int u = 0; u < s; ++u) n += rd_n(u, S, NP);
  return n;
}
constexpr int rd_ord_b(int j, int S, int NP) { return j == 0 ? 2 : (j == 1 ? 3 : rd_before(j - 2, S, NP) + 1); }
constexpr int rd_ord_a(int p, int S, int NP) { return p == 0 ? 1 : rd_before(4 * p - 3, S, NP) + (4 * p - 1 < S ? 1 : 0) + 1; }
constexpr int rd_wait(int s, int S, int NP) {
  const int b = rd_ord_b(s, S, NP), a = rd_ord_a(s / 4, S, NP);
  return rd_before(s, S, NP) - (a > b ? a : b);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int TM = 128, TN = 128;

struct Gemm2Args {
  ConvGemmBatch batch;
  int first[kMaxGemmBatch];   // first linear tile of each problem
  int mt[kMaxGemmBatch];      // m-tiles of each problem (m runs fastest inside a problem)
  int xcd_map;                // 1: XCD-aware tile order (below); 2: conv-bank order (problems dealt to XCDs, below)
  // conv-bank order: XCD x works on tiles [a0, a0 + na) of problem pa, then on tiles [b0, b0 + nb) of problem pb
  int bk_pa[8], bk_a0[8], bk_na[8], bk_pb[8], bk_b0[8], bk_nb[8];
};

// BX: 0 = v_mfma_f32_32x32x2_f32; 1 = bf16x3 products, both operands split into planes in registers; 2 (round 6) = bf16x3 with
// the B (weight) operand read as READY-MADE plane fragments from a pre-split image (WImg below): only A is split in registers
template <int BK, int NS, int BX = 0>
__global__ __launch_bounds__(256, 2) void conv_gemm2_kernel(Gemm2Args G) {
  static_assert(!BX || BK == 16, "the bf16x3 form works on 16-deep k-tiles (one v_mfma_f32_32x32x16_bf16 group per tile)");
  constexpr int SLOTS = BK / 4;               // 16-byte slots per A row
  constexpr int A_RPI = 64 / SLOTS;           // A rows per wave instruction: 8 (BK = 32) / 16 (BK = 16)
  constexpr int A_INSTR = TM / A_RPI / 4;     // per wave
  constexpr int B_INSTR = BX == 2 ? kWImgTileBytes / 4096 : BK / 2 / 4;   // one instruction = 2 k-rows of 128 floats (BX == 2: 1 KB of the image tile); per wave
  constexpr int NLD = A_INSTR + B_INSTR;      // DMA instructions per wave per k-tile
  constexpr int A_FLOATS = TM * BK, B_FLOATS = BX == 2 ? kWImgTileBytes / 4 : BK * TN, STAGE = A_FLOATS + B_FLOATS;
  constexpr int SWZ_SH = BK == 32 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // NS * STAGE floats, the ONLY LDS object of the kernel

  // Tile order.  The dispatcher deals workgroups to the 8 XCDs round-robin (workgroup id % 8), each XCD with its own L2.
  //  * xcd_map (batches whose tiles all run equally long): XCD x takes a CONTIGUOUS eighth of the linear tile list, in which
  //    the n-tiles of one m-tile are neighbours -- they are resident together in ONE L2: the A rows they share are fetched from
  //    the fabric once (not once per XCD the old order spread them over), and the partial cache lines two neighbouring
  //    n-tiles write at their seam (row pitch 1025) merge in that L2.
  //  * otherwise (conv banks: problems of different depth, longest first) the list is dealt in order so that every XCD gets
  //    the same mix of long and short tiles.
  //  * xcd_map == 2 (round 5; conv banks of 16 or 8 widths with equal tile counts): whole PROBLEMS are dealt to the XCDs -- the
  //    widths are paired deepest with shallowest (16 + 1, 15 + 2, ... taps: every pair is the same work), one pair per XCD (16
  //    widths) or per two XCDs that split its m-tiles (8 widths).  An XCD then streams the tap weights of its own two widths only
  //    (17 x 64 KB for the encoder bank: L2-resident) instead of all 8.9 MB per m-tile.
  int lin = blockIdx.x;
  int bank_pi = -1, bank_rel = 0;
  if (G.xcd_map == 2) {
    const int x = lin & 7, j = lin >> 3;
    if (j < G.bk_na[x]) { bank_pi = G.bk_pa[x]; bank_rel = G.bk_a0[x] + j; }
    else if (j < G.bk_na[x] + G.bk_nb[x]) { bank_pi = G.bk_pb[x]; bank_rel = G.bk_b0[x] + j - G.bk_na[x]; }
    else return;   // (padding of the shorter halves)
  }
  if (G.xcd_map == 1) {
    const int tiles = gridDim.x, q = tiles >> 3, rem = tiles & 7, x = lin & 7, j = lin >> 3;
    lin = x * q + (x < rem ? x : rem) + j;
  }
  int pi = 0;
  if (bank_pi >= 0) {
    pi = bank_pi;
  } else {
    for (int i = 1; i < G.batch.n; ++i)
      if (lin >= G.first[i]) pi = i;
  }
  const ConvGemmProblem& P = G.batch.p[pi];
  const int rel = bank_pi >= 0 ? bank_rel : lin - G.first[pi];
  const int mtiles = G.mt[pi];
  int tnn, tmm;
  if (G.xcd_map == 1) {
    const int ntiles = (P.N + TN - 1) / TN;
    tmm = rel / ntiles;
    tnn = rel - tmm * ntiles;
  } else {
    tnn = rel / mtiles;
    tmm = rel - tnn * mtiles;
  }
  const int m0 = tmm * (P.pool ? TM - 1 : TM), n0 = tnn * TN;   // pooled epilogue: tiles overlap by their halo row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS destinations (M0) stay on the scalar unit
  const int T = P.T, K = P.K, lda = P.lda, ldw = P.ldw;
  const int pad_l = P.pad_l;

  // ---- DMA source setup: each thread serves the same A rows / B slots for every k-tile.  Byte offsets against
  //      A - pad_l rows (so that the tap shift is a non-negative scalar) and W; the launcher bounds both below 2 GiB ----
  const __amdgpu_buffer_rsrc_t rsA = dma_rsrc(P.A - (int64_t)pad_l * lda),
                               rsW = dma_rsrc(BX == 2 ? reinterpret_cast<const float*>(P.Wimg) : P.W);
  int a_vo[A_INSTR], a_cur[A_INSTR], a_t[A_INSTR], a_klim[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * A_RPI + lane / SLOTS;
    const int q = (lane % SLOTS) ^ ((row >> SWZ_SH) & (SLOTS - 1));   // which logical slot lands in this lane's LDS slot
    const int m = m0 + row;
    const bool ok = m < P.M;
    a_t[i] = ok ? m % T : -(1 << 28);
    a_klim[i] = K - 4 * q;                       // the slot is inside the row while k0 < a_klim
    a_vo[i] = ((ok ? m : 0) * lda + 4 * q) * 4;
  }
  const int b_c = n0 + 4 * (lane & 31);
  const bool b_ok = b_c < P.Nld;
  const int b_r0 = wave * B_INSTR * 2 + (lane >> 5);
  const int b_klim = K - b_r0;                   // row k0 + 2 i + b_r0 exists while k0 + 2 i < b_klim
  // BX == 2: the image tile of (n-tile, k-tile `it`) is 12 KB of fragment-ordered planes, contiguous, already zero-padded:
  // wave w copies its 1 KB pieces 4 i + w, no masks; the tile index goes into the scalar offset
  const int b_vo = BX == 2 ? (wave * 1024 + lane * 16) : (b_ok ? (b_r0 * ldw + b_c) * 4 : kOOB);

  // k-tiles per tap, counted so that a tap always spans whole 32-deep units (a k-split chunk [it0, it1) is given in those
  // units whatever BK is; with BK = 16 an odd tail tile is all-masked zeros)
  const int ktiles = ((K + 31) / 32) * (32 / BK);
  const int it_begin = P.it0 * (32 / BK), it_end = (P.it1 > 0 ? P.it1 * (32 / BK) : P.taps * ktiles);
  const int nit = it_end - it_begin;

  // next tile to issue: (tap, k0) advance incrementally (no division on the loop path)
  int n_tap = it_begin / ktiles, n_k0 = (it_begin - n_tap * ktiles) * BK;
  int n_img = (tnn * P.img_its + it_begin) * kWImgTileBytes;   // BX == 2: byte offset of the next image tile to issue
  // row shift of the tap (t_sh) and, in the conv bank's gather mode (ConvGemmProblem::bank_filters), the filter it belongs to
  // (bf, tap bj of it) with the column block of A that filter reads (t_col)
  const int bankF = P.bank_filters;
  int bf = 1, bj = n_tap, t_sh = n_tap - pad_l, t_col = 0;
  if (bankF) {
    while (bj >= bf) {
      bj -= bf;
      ++bf;
    }
    t_sh = bj - ((bf - 1) - (bf - 1) / 2);
    t_col = (bf - 1) * K;
  }
  // rows a shifted tap pulls from outside their sequence are masked for the whole tap: recomputed when the tap changes
  auto retap = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) a_cur[i] = (unsigned)(a_t[i] + t_sh) < (unsigned)T ? a_vo[i] : kOOB;
  };
  retap();
  // one DMA instruction of the next tile (g < A_INSTR: A rows, else B rows) -- interleaved between the MFMA groups below.
  // Per piece: one compare + select for the K tail (never taken when K is a multiple of BK), scalar offsets, the load.
  auto issue_piece = [&](int g, int stage) __attribute__((always_inline)) {
    float* As = smem + stage * STAGE;
    if (g < A_INSTR) {
      const int vo = n_k0 < a_klim[g] ? a_cur[g] : kOOB;
      blds16(rsA, vo, ((t_sh + pad_l) * lda + t_col + n_k0) * 4, As + (wave * A_INSTR + g) * A_RPI * BK);
    } else {
      const int i = g - A_INSTR;
      if constexpr (BX == 2) {
        blds16(rsW, b_vo, n_img + i * 4096, As + A_FLOATS + (4 * i + wave) * 256);
      } else {
        const int vo = n_k0 + 2 * i < b_klim ? b_vo : kOOB;
        blds16(rsW, vo, ((n_tap * K + n_k0 + 2 * i) * ldw) * 4, As + A_FLOATS + (wave * B_INSTR + i) * 2 * TN);
      }
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    if constexpr (BX == 2) n_img += kWImgTileBytes;
    n_k0 += BK;
    if (n_k0 >= ktiles * BK) {
      n_k0 = 0;
      ++n_tap;
      if (bankF) {
        if (++bj == bf) {
          bj = 0;
          ++bf;
        }
        t_sh = bj - ((bf - 1) - (bf - 1) / 2);
        t_col = (bf - 1) * K;
      } else {
        t_sh = n_tap - pad_l;
        t_col = 0;   // (both branches assign both variables: with one store per branch hipcc sinks them into ONE store through a
                     //  phi of the two stack addresses, which keeps t_sh / t_col in scratch memory -- two scratch loads and an
                     //  `s_waitcnt vmcnt(0)`, i.e. a drain of the whole DMA ring, in every k-tile of rounds 2-4's kernel)
      }
      retap();
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#ifndef TACO_BF16X_ACC1
  f32x16 acc_lo[BX ? 4 : 1];   // bf16x3: the five low-order plane products of every sub-tile (bf16x3.h mfma6_2); added to acc behind the k-loop
#pragma unroll
  for (int j = 0; j < (BX ? 4 : 1); ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_lo[j][e] = 0.f;
#define MF6(j, a, b) mfma6_2(acc[j], acc_lo[j], a, b)
#else
#define MF6(j, a, b) mfma6(acc[j], a, b)
#endif

  const int li = lane & 31, kh = lane >> 5;
  const int arow = wave * 32 + li;
  const int aswz = (arow >> SWZ_SH) & (SLOTS - 1);
  constexpr int NP = BK / 8, NSTEP = NP * 4;
  // per-lane LDS byte offsets inside a stage: A slot of step p (swizzled), B row base
  uint32_t a_off[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) a_off[p] = (uint32_t)(arow * BK + 4 * ((2 * p + kh) ^ aswz)) * 4u;
  const uint32_t b_off = (uint32_t)(A_FLOATS + 4 * kh * TN + 4 * li) * 4u;
  const uint32_t lds0 = lds_off(smem);

  // 4 NP steps of four MFMAs per tile.  Fragments are requested two steps (B) / three steps (A) ahead with counted
  // lgkmcnt waits; one DMA instruction of the tile after next is issued in the shadow of each of the first NLD MFMA groups.
  auto compute = [&](int stage, int fill_stage, auto fill_c) __attribute__((always_inline)) {
    constexpr bool fill = decltype(fill_c)::value;
    const uint32_t sb = lds0 + (uint32_t)stage * (STAGE * 4u);
    const uint32_t bb = sb + b_off;
    f32x4 a4[2], b4[3];
    a4[0] = dsr128<0>(sb + a_off[0]);
    b4[0] = dsr128<0>(bb);
    b4[1] = dsr128<TN * 4>(bb);
    static_for<0, NSTEP>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int p = s >> 2, c = s & 3;
      wait_lgkm<rd_wait(s, NSTEP, NP)>();
      __builtin_amdgcn_sched_barrier(0);
      const float av = a4[p & 1][c];
      const f32x4 bv = b4[s % 3];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[3], acc[3], 0, 0, 0);
      // (no pin between the MFMAs and what follows: the scheduler spreads the reads and the DMA set-up over the MFMA shadows;
      //  the only fixed point is the wait + sched_barrier at the head of the next step -- guide rule 18)
      if (s + 2 < NSTEP) b4[(s + 2) % 3] = dsr128<(8 * ((s + 2) >> 2) + ((s + 2) & 3)) * TN * 4>(bb);
      if (c == 1 && p + 1 < NP) a4[(p + 1) & 1] = dsr128<0>(sb + a_off[p + 1]);
#ifndef GEMM2_LAB_NODMA
      if (s < NLD && fill) issue_piece(s, fill_stage);
#endif
    });
  };

  // ---- the same tile on the bf16 matrix pipe (BX, BK = 16).  The lane's 8 k-slots are q = 4 p + c <-> k = 8 p + 4 kh + c,
  //      the SAME ten ds_read_b128 as above; A and every B sub-tile column are split into three bf16 planes in registers
  //      (split8) and sub-tile j takes six MFMAs.  Software pipeline across the barrier: the reads of tile t are issued first;
  //      sub-tile 2 of tile t - 1 (planes carried in registers) and the DMA issue of the tile after next cover their LDS
  //      latency; sub-tile 3 of tile t - 1 runs beside the split of A and B0, sub-tile 0 beside the split of B1, sub-tile 1
  //      beside B2 / B3.  (First tile: the carried planes are zeros.) ----
  Pl3 pa_c = zero_pl3(), pb2_c = zero_pl3(), pb3_c = zero_pl3();
  auto compute_bx = [&](int stage, int fill_stage, auto fill_c) __attribute__((always_inline)) {
    constexpr bool fill = decltype(fill_c)::value;
    const uint32_t sb = lds0 + (uint32_t)stage * (STAGE * 4u);
    const uint32_t bb = sb + b_off;
    const f32x4 ra0 = dsr128<0>(sb + a_off[0]);
    const f32x4 ra1 = dsr128<0>(sb + a_off[NP - 1]);
    f32x4 rb[8];
    static_for<0, 8>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      rb[q] = dsr128<(8 * (q >> 2) + (q & 3)) * TN * 4>(bb);
    });
    __builtin_amdgcn_sched_barrier(0);
    MF6(2, pa_c, pb2_c);
#ifndef GEMM2_LAB_NODMA
    if (fill) {
#pragma unroll
      for (int g = 0; g < NLD; ++g) issue_piece(g, fill_stage);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    MF6(3, pa_c, pb3_c);   // (still the previous tile's A planes: pa_n takes over below)
    const Pl3 pa_n = split8(ra0[0], ra0[1], ra0[2], ra0[3], ra1[0], ra1[1], ra1[2], ra1[3]);
    const Pl3 p0 = split8(rb[0][0], rb[1][0], rb[2][0], rb[3][0], rb[4][0], rb[5][0], rb[6][0], rb[7][0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 15, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    pa_c = pa_n;
    MF6(0, pa_c, p0);
    const Pl3 p1 = split8(rb[0][1], rb[1][1], rb[2][1], rb[3][1], rb[4][1], rb[5][1], rb[6][1], rb[7][1]);
    __builtin_amdgcn_sched_barrier(0);
    MF6(1, pa_c, p1);
    pb2_c = split8(rb[0][2], rb[1][2], rb[2][2], rb[3][2], rb[4][2], rb[5][2], rb[6][2], rb[7][2]);
    pb3_c = split8(rb[0][3], rb[1][3], rb[2][3], rb[3][3], rb[4][3], rb[5][3], rb[6][3], rb[7][3]);
    // one MFMA, then its share of the 88 split instructions (left alone hipcc issues the six MFMAs back to back and sinks the
    // splits behind the loop branch, where nothing of this wave covers them)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 15, 0);
    }
    // (the carried planes are pinned HERE: they are only read in the next iteration, and machine sinking would move their
    //  computation into the block behind the loop branch)
    asm volatile("" : "+v"(pb2_c.h), "+v"(pb2_c.m), "+v"(pb2_c.l), "+v"(pb3_c.h), "+v"(pb3_c.m), "+v"(pb3_c.l));
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- BX == 2: the B planes come ready-made from the image (12 conflict-free ds_read_b128: plane P of sub-tile j is the
  //      wave's 1 KB run (4 P + j), lane-ordered), only A is split: 44 instead of 220 VALU operations per 24 MFMAs.  Same software
  //      pipeline as above: sub-tiles 2 / 3 of tile t - 1 (planes carried in registers) cover the LDS latency of tile t's reads. ----
  Pl3 qb2_c = zero_pl3(), qb3_c = zero_pl3();
  auto compute_bi = [&](int stage, int fill_stage, auto fill_c) __attribute__((always_inline)) {
    constexpr bool fill = decltype(fill_c)::value;
    const uint32_t sb = lds0 + (uint32_t)stage * (STAGE * 4u);
    const uint32_t bb = sb + (uint32_t)(A_FLOATS * 4) + (uint32_t)lane * 16u;
    const f32x4 ra0 = dsr128<0>(sb + a_off[0]);
    const f32x4 ra1 = dsr128<0>(sb + a_off[NP - 1]);
    u32x4 pl[4][3];
    static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      static_for<0, 3>([&](auto pc) {
        constexpr int pp = decltype(pc)::value;
#ifdef GEMM2_LAB_NOBREAD   // timing lab: no B fragment reads (results are garbage)
        pl[j][pp] = u32x4{(unsigned)bb, 0x3f803f80u, (unsigned)j, (unsigned)pp};
#else
        pl[j][pp] = __builtin_bit_cast(u32x4, dsr128<(4 * pp + j) * 1024>(bb));
#endif
      });
    });
    __builtin_amdgcn_sched_barrier(0);
#if !defined(TACO_BF16X_ACC1) && !defined(TACO_GEMM2_NO_PAIR23)
    // Sub-tiles 2 and 3 INTERLEAVED (round 6, late): as two blocks of six, five MFMAs of each block were back-to-back on ONE accumulator
    // with the DMA issues / the split's VALU between them -- an instruction between two MFMAs on the same accumulator costs ~40
    // cycles, between MFMAs on different accumulators ~6 (MI355X guide).  Every accumulator still receives its products in mfma6_2's
    // order: results are bit-identical.  -DTACO_GEMM2_NO_PAIR23: the previous form.
    acc_lo[2] = mfma_bf(pa_c.l, qb2_c.h, acc_lo[2]);
    acc_lo[3] = mfma_bf(pa_c.l, qb3_c.h, acc_lo[3]);
    acc[2] = mfma_bf(pa_c.h, qb2_c.h, acc[2]);
    acc[3] = mfma_bf(pa_c.h, qb3_c.h, acc[3]);
    acc_lo[2] = mfma_bf(pa_c.h, qb2_c.l, acc_lo[2]);
    acc_lo[3] = mfma_bf(pa_c.h, qb3_c.l, acc_lo[3]);
#else
    MF6(2, pa_c, qb2_c);
#endif
#ifndef GEMM2_LAB_NODMA
    if (fill) {
#pragma unroll
      for (int g = 0; g < NLD; ++g) issue_piece(g, fill_stage);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifdef GEMM2_LAB_NOBREAD
    wait_lgkm<0>();
#else
    wait_lgkm<12>();                 // the two A reads were issued first
#endif
    __builtin_amdgcn_sched_barrier(0);
#if !defined(TACO_BF16X_ACC1) && !defined(TACO_GEMM2_NO_PAIR23)
    acc_lo[2] = mfma_bf(pa_c.m, qb2_c.m, acc_lo[2]);      // (still the previous tile's A planes)
    acc_lo[3] = mfma_bf(pa_c.m, qb3_c.m, acc_lo[3]);
    acc_lo[2] = mfma_bf(pa_c.m, qb2_c.h, acc_lo[2]);
    acc_lo[3] = mfma_bf(pa_c.m, qb3_c.h, acc_lo[3]);
    acc_lo[2] = mfma_bf(pa_c.h, qb2_c.m, acc_lo[2]);
    acc_lo[3] = mfma_bf(pa_c.h, qb3_c.m, acc_lo[3]);
#else
    MF6(3, pa_c, qb3_c);      // (still the previous tile's A planes)
#endif
    const Pl3 pa_n = split8(ra0[0], ra0[1], ra0[2], ra0[3], ra1[0], ra1[1], ra1[2], ra1[3]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {    // one MFMA, then its share of the 44 split instructions
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    pa_c = pa_n;
    Pl3 q0, q1;
    q0.h = pl[0][0]; q0.m = pl[0][1]; q0.l = pl[0][2];
    q1.h = pl[1][0]; q1.m = pl[1][1]; q1.l = pl[1][2];
    MF6(0, pa_c, q0);
    MF6(1, pa_c, q1);
    qb2_c.h = pl[2][0]; qb2_c.m = pl[2][1]; qb2_c.l = pl[2][2];
    qb3_c.h = pl[3][0]; qb3_c.m = pl[3][1]; qb3_c.l = pl[3][2];
    asm volatile("" : "+v"(qb2_c.h), "+v"(qb2_c.m), "+v"(qb2_c.l), "+v"(qb3_c.h), "+v"(qb3_c.m), "+v"(qb3_c.l));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run_tile = [&](int stage, int fill_stage, auto fill_c) __attribute__((always_inline)) {
    if constexpr (BX == 2) compute_bi(stage, fill_stage, fill_c);
    else if constexpr (BX == 1) compute_bx(stage, fill_stage, fill_c);
    else compute(stage, fill_stage, fill_c);
  };

  // ---- NS-stage ring: tile t lives in stage t % NS; tiles up to t + NS - 2 are in flight while t is computed ----
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nit) {
#pragma unroll
      for (int g = 0; g < NLD; ++g) issue_piece(g, s);
      advance();
    }
  // Every scalar the loop body reads is consumed once HERE, so the compiler waits for the kernarg loads in front of the loop
  // instead of inserting `s_waitcnt lgkmcnt(0)` at their first use inside it (which would also drain the fragment reads);
  // from here on lgkmcnt counts the LDS reads below and nothing else.
  asm volatile("" ::"s"(T), "s"(K), "s"(lda), "s"(ldw), "s"(pad_l), "s"(nit), "s"(n_tap), "s"(n_k0), "s"(ktiles));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int it = 0;
  for (; it + NS - 1 < nit; ++it) {   // steady state: tile it + NS - 1 refills the stage tile it - 1 just vacated
    // this wave's share of tile `it` has landed once at most the younger tiles' DMA instructions are outstanding
    wait_vm<NLD*(NS - 2)>();
#ifndef GEMM2_LAB_NOBAR
    __builtin_amdgcn_s_barrier();   // every wave's share has landed AND every wave has finished reading tile it - 1
#endif
    asm volatile("" ::: "memory");
    run_tile(it % NS, (it + NS - 1) % NS, std::true_type{});
    advance();
  }
  for (; it < nit; ++it) {            // drain: nothing left to request
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    run_tile(it % NS, 0, std::false_type{});
  }
  if constexpr (BX == 1) {   // the last tile's sub-tiles 2 / 3
    MF6(2, pa_c, pb2_c);
    MF6(3, pa_c, pb3_c);
  }
  if constexpr (BX == 2) {
    MF6(2, pa_c, qb2_c);
    MF6(3, pa_c, qb3_c);
  }
#ifndef TACO_BF16X_ACC1
  if constexpr (BX != 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += acc_lo[j];
  }
#endif
#undef MF6
  wait_vm<0>();   // (nothing outstanding by construction; keeps the invariant explicit before the epilogue's ordinary loads)

  // ---- epilogue.  C/D layout of v_mfma_f32_32x32x2: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5);
  //      sub-tile j holds columns n0 + 4 li + j, so element e of the four accumulators is one float4 of row `row` ----
  const int n = n0 + 4 * li;
  if (P.pool == 2) {
    // ---- backward pooled epilogue (kernels.h: ConvGemmProblem::pool == 2).  The accumulators hold dy = d(pooled) of rows
    //      m0 .. m0+127; row m needs dy[m-1] (same lane, the lane 32 away, or the wave below via LDS -- the mirror image of the
    //      forward epilogue) and the stored activations x[m-1], x[m], x[m+1].  Three passes per half of the tile's rows: all
    //      loads, all arithmetic, all stores. ----
    const bool col_ok = n + 3 < P.N;
    float sc[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[j] = P.scale[n + j] * P.scale_mul;
        be[j] = P.shift[n + j];
      }
    }
    // dy of the previous row for the first row of every 4-row group
    __syncthreads();   // the ring is free: its first 2 KB carry each wave's row 31 to the wave above
    if (kh == 1) *reinterpret_cast<float4*>(smem + wave * TN + 4 * li) = make_float4(acc[0][15], acc[1][15], acc[2][15], acc[3][15]);
    __syncthreads();
    float pv[4][4];   // pv[j][i]: dy of the row in front of this lane's row e = 4 i
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float got = __shfl_xor(acc[j][4 * i + 3], 32);   // rows 8i+3 (from kh = 0) / 8i+7 (from kh = 1)
        if (kh == 1) pv[j][i] = got;                            // row 8i+4 follows row 8i+3
        else if (i < 3) pv[j][i + 1] = got;                     // row 8(i+1) follows row 8i+7
      }
    if (kh == 0) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wave > 0) t = *reinterpret_cast<const float4*>(smem + (wave - 1) * TN + 4 * li);
      pv[0][0] = t.x; pv[1][0] = t.y; pv[2][0] = t.z; pv[3][0] = t.w;
    }
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    const float rs = P.scale_mul;
    const float* X = P.pool_x + n;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      // pass 1: x rows r0-1 .. r0+4 of the two 4-row groups of this half (clamped addresses; masked by the predicates below)
      f32x4 xr[2][6];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int r0 = m0 + wave * 32 + 8 * (2 * hf + g) + 4 * kh;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          int m = r0 - 1 + q;
          m = m < 0 ? 0 : (m >= P.M ? P.M - 1 : m);
          xr[g][q] = col_ok ? *reinterpret_cast<const f32x4*>(X + (int64_t)m * P.ldc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      // pass 2
      f32x4 outv[2][4];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 2 * hf + g, e = 4 * i + q;
          const int lr = wave * 32 + 8 * i + 4 * kh + q;   // local row
          const int m = m0 + lr;
          const int t = m % T;
          const bool own = (lr >= 1 || tmm == 0) && m < P.M;
          const bool first = t == 0, last = t == T - 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xc = xr[g][q + 1][j], xp = xr[g][q][j], xn = xr[g][q + 2][j];
            const float z = xc * sc[j] + be[j];
            const float dyc = acc[j][e];
            const float dyp = q > 0 ? acc[j][e - 1] : pv[j][i];
            float dz = 0.f;
            if (last || z >= xn * sc[j] + be[j]) dz = dyc;
            if (!first && z > xp * sc[j] + be[j]) dz += dyp;
            if (!own) dz = 0.f;
            outv[g][q][j] = xc > 0.f ? dz * sc[j] : 0.f;
            ag[j] += dz * xc * rs;
            ab[j] += dz;
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      // pass 3
      if (col_ok) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int lr = wave * 32 + 8 * (2 * hf + g) + 4 * kh + q;
            const int m = m0 + lr;
            if ((lr >= 1 || tmm == 0) && m < P.M) *reinterpret_cast<f32x4*>(P.C + (int64_t)m * P.ldc + n) = outv[g][q];
          }
      }
    }
    // column sums: the two row halves of a wave, then the four waves (fixed order inside the workgroup), one atomic per column
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ag[j] += __shfl_xor(ag[j], 32);
      ab[j] += __shfl_xor(ab[j], 32);
    }
    __syncthreads();
    if (kh == 0) {
      *reinterpret_cast<float4*>(smem + (2 * wave) * TN + 4 * li) = make_float4(ag[0], ag[1], ag[2], ag[3]);
      *reinterpret_cast<float4*>(smem + (2 * wave + 1) * TN + 4 * li) = make_float4(ab[0], ab[1], ab[2], ab[3]);
    }
    __syncthreads();
    if (tid < 2 * TN) {
      const int which = tid / TN, c = tid - which * TN;
      const float v = smem[(0 + which) * TN + c] + smem[(2 + which) * TN + c] + smem[(4 + which) * TN + c] + smem[(6 + which) * TN + c];
      if (n0 + c < P.N) atomicAdd((which ? P.pool_dbeta : P.pool_dgamma) + n0 + c, v);
    }
    return;
  }
  if (P.pool) {
    // ---- pooled epilogue (conv bank, ops.py:62-71): bank = act(acc + bias) -> Cpre;  z = bank * scale + shift;
    //      C[m] = max(z[m], z[m + 1]) inside a sequence.  Row m + 1 of element e lives in the same lane (e & 3 < 3), in the
    //      lane 32 away (rows 3|4 and 7|8 of an 8-row group straddle the kh halves) or in the next wave (row 31|32, via LDS);
    //      local row 127 is the halo: it only feeds row 126 unless it ends its sequence (the next tile starts ON it).
    //      Launcher contract: float4 rows, no keep / residual / per-sequence bias / atomics. ----
    const bool col_ok = n + 3 < P.N;
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sf[4] = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (P.bias) b4[j] = P.bias[n + j];
        if (P.scale) sc[j] = P.scale[n + j] * P.scale_mul;
        if (P.shift) sf[j] = P.shift[n + j];
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = apply_act(acc[j][e] + b4[j], P.act);
      if (P.Cpre && col_ok && m < P.M)
        *reinterpret_cast<float4*>(P.Cpre + (int64_t)m * P.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j][e] = v[j] * sc[j] + sf[j];
    }
    __syncthreads();   // every wave is done with the ring: its first 2 KB now carry each wave's row 0 to the wave above
    if (kh == 0) *reinterpret_cast<float4*>(smem + wave * TN + 4 * li) = make_float4(acc[0][0], acc[1][0], acc[2][0], acc[3][0]);
    __syncthreads();
    float nx[4][4];   // nx[j][i]: successor of this lane's row (e = 4 i + 3)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float send = kh ? acc[j][4 * i] : acc[j][i < 3 ? 4 * (i + 1) : 12];
        nx[j][i] = __shfl_xor(send, 32);
      }
    if (kh == 1 && wave < 3) {
      const float4 t = *reinterpret_cast<const float4*>(smem + (wave + 1) * TN + 4 * li);
      nx[0][3] = t.x; nx[1][3] = t.y; nx[2][3] = t.z; nx[3][3] = t.w;
    }
    if (!col_ok) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      if (m >= P.M) continue;
      const bool has_next = (m % T) + 1 < T;
      if (has_next && wave == 3 && kh == 1 && e == 15) continue;   // halo row: written by the tile that starts on it
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float nxt = (e & 3) < 3 ? acc[j][(e + 1) & 15] : nx[j][e >> 2];
        o[j] = has_next ? fmaxf(acc[j][e], nxt) : acc[j][e];
      }
      *reinterpret_cast<float4*>(P.C + (int64_t)m * P.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  if (n >= P.N) return;
  const bool vec = (P.flags & 4) != 0 && n + 3 < P.N;
  float bias0[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (n + j < P.N) {
      if (P.bias && P.bias_stride == 0) bias0[j] = P.bias[n + j];
      if (P.scale) sc[j] = P.scale[n + j] * P.scale_mul;
      if (P.shift) sf[j] = P.shift[n + j];
    }
  }
  const bool affine = P.scale || P.shift;
  if (P.flags & 8) {
    // ---- rows whose pitch is not a multiple of 4 floats (the final dense layer: 1025 columns, tacotron.py:148).  Element
    //      (m, c) of a 16-byte aligned C sits at float index m * ldc + c, so the aligned float4 groups of row m start at
    //      columns c = s (mod 4) with s = (-m * ldc) mod 4 -- the same for every lane of the wave (rows differ by multiples of 4
    //      between lanes).  A lane owns columns n .. n+3; it stores the ALIGNED group n+s .. n+s+3 = its own elements s..3 and the
    //      first s elements of the lane to its right (DPP wave_shl:1), lane 0 adds the s head columns of the tile as scalars and
    //      lane 31 its last 4-s (the group beyond belongs to the next tile): 16 float4 stores per lane instead of 64 scalar ones.
    //      Launcher contract (bit 3): bias / activation / affine only.
    const int ldm = P.ldc & 3;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      float v[4], nx[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = apply_act(acc[j][e] + bias0[j], P.act);
        if (affine) x = x * sc[j] + sf[j];
        v[j] = x;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) nx[j] = dpp_move<0x130>(0.f, v[j]);   // wave_shl:1 -- lane i <- lane i + 1
      if (m >= P.M) continue;
      // rows e, e+4, ... of the two halves share m mod 4; the shift is wave-uniform per e
      const int sft = (4 - (((m0 + (e & 3)) & 3) * ldm & 3)) & 3;
      float* crow = P.C + (int64_t)m * P.ldc;
      float o[4];
      switch (sft) {
        case 0: o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; break;
        case 1: o[0] = v[1]; o[1] = v[2]; o[2] = v[3]; o[3] = nx[0]; break;
        case 2: o[0] = v[2]; o[1] = v[3]; o[2] = nx[0]; o[3] = nx[1]; break;
        default: o[0] = v[3]; o[1] = nx[0]; o[2] = nx[1]; o[3] = nx[2]; break;
      }
      const int c0 = n + sft;
      if (li < 31 && c0 + 3 < P.N) {
        *reinterpret_cast<float4*>(crow + c0) = make_float4(o[0], o[1], o[2], o[3]);
      } else {   // the tile's last lane / the N tail: this lane's own elements one by one, and (inside the tile) the elements of
                 // the lane to the right that its own group would have carried
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j >= sft && n + j < P.N) crow[n + j] = v[j];
        if (li < 31) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (j < sft && n + 4 + j < P.N) crow[n + 4 + j] = nx[j];
        }
      }
      if (li == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < sft && n + j < P.N) crow[n + j] = v[j];
      }
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
    if (m >= P.M) continue;
    float v[4] = {acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
    const float* brow = P.bias_stride ? P.bias + (int64_t)(m / T) * P.bias_stride + n : nullptr;
    if (vec) {
      uint32_t kp = 0x01010101u;
      if (P.keep) kp = *reinterpret_cast<const uint32_t*>(P.keep + (int64_t)m * P.N + n);
      float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.residual) res = *reinterpret_cast<const float4*>(P.residual + (int64_t)m * P.ldr + n);
      const float rr[4] = {res.x, res.y, res.z, res.w};
      float pre[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = apply_act(v[j] + (brow ? brow[j] : bias0[j]), P.act);
        if (P.keep) x = ((kp >> (8 * j)) & 0xffu) ? x * 2.0f : 0.0f;
        pre[j] = x;
        if (affine) x = x * sc[j] + sf[j];
        v[j] = x + rr[j];
      }
      if (P.Cpre) *reinterpret_cast<float4*>(P.Cpre + (int64_t)m * P.ldc + n) = make_float4(pre[0], pre[1], pre[2], pre[3]);
      float* c = P.C + (int64_t)m * P.ldc + n;
      if (P.atomic_out) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(c + j, v[j]);
      } else {
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (n + j >= P.N) continue;
        float x = apply_act(v[j] + (brow ? brow[j] : bias0[j]), P.act);
        if (P.keep) x = P.keep[(int64_t)m * P.N + n + j] ? x * 2.0f : 0.0f;
        if (P.Cpre) P.Cpre[(int64_t)m * P.ldc + n + j] = x;
        if (affine) x = x * sc[j] + sf[j];
        if (P.residual) x += P.residual[(int64_t)m * P.ldr + n + j];
        if (P.atomic_out) atomicAdd(&P.C[(int64_t)m * P.ldc + n + j], x);
        else P.C[(int64_t)m * P.ldc + n + j] = x;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM, second generation:  dW[tap][k][n] += sum_m A[row(m, tap)][k] * dY[m][n]   (tacotron.py:172)
//
// Same machinery as conv_gemm2_kernel with the REDUCTION running over rows: a stage holds 32 rows of A (128 k-columns each)
// and the same 32 rows of dY (128 n-columns), both as plain row images (512 B per row, DMA-written, no swizzle needed: every
// fragment read is a run of consecutive 8-byte words).  Wave (wm, wn) owns the 64 x 64 block of the 128 x 128 output tile;
// one ds_read_b64 per operand feeds four MFMAs: lane (i, kh) holds A[m = 2s + kh][64 wm + 2i + {0,1}] and
// dY[m][64 wn + 2i + {0,1}], i.e. two interleaved 32-wide sub-tiles per operand.  The row range of a problem is split over
// `splits` workgroups that combine with fp32 atomics (as gemm.hip does); bias gradients (column sums of dY) are taken from the
// LDS image by the k-tile-0 / tap-0 workgroups.
struct Tn2Args {
  GemmTnArgs p[kMaxTnBatch];
  int first[kMaxTnBatch], gx[kMaxTnBatch], gy[kMaxTnBatch];
  int n;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ f32x2 dsr64(uint32_t addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

__global__ __launch_bounds__(256, 2) void gemm_tn2_kernel(Tn2Args G) {
  constexpr int BR = 32;                      // rows (reduction depth) per stage
  constexpr int OP_FLOATS = BR * 128, STAGE = 2 * OP_FLOATS;
  constexpr int NLD = 8;                      // 4 + 4 DMA instructions per wave per stage (one instruction = 2 rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages

  int pi = 0;
  for (int i = 1; i < G.n; ++i)
    if ((int)blockIdx.x >= G.first[i]) pi = i;
  const GemmTnArgs& P = G.p[pi];
  int rel = blockIdx.x - G.first[pi];
  const int gx = G.gx[pi], gy = G.gy[pi];
  int bz = rel / (gx * gy);
  rel -= bz * gx * gy;
  const int by = rel / gx, bx = rel - by * gx;
  const int split = bz % P.splits;
  const int tap = bz / P.splits;
  const int k0 = bx * 128, n0 = by * 128;
  const int K = P.K, N = P.N, T = P.T, lda = P.lda, ldy = P.ldy;
  const int sh = tap - P.pad_l;
  const int m_begin = split * P.chunk;
  const int m_end = min(P.M, m_begin + P.chunk);
  const int nit = (m_end - m_begin + BR - 1) / BR;
  if (nit <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;
  const float* zero = g_zero4;
  asm volatile("" : "+v"(zero));
  auto pick = [&](const float* p, bool ok) {
    const uint64_t m = ok ? ~0ull : 0ull;
    return reinterpret_cast<const float*>((reinterpret_cast<uint64_t>(p) & m) | (reinterpret_cast<uint64_t>(zero) & ~m));
  };

  // DMA sources: this thread serves rows (wave * 4 + i) * 2 + kh of every stage, 16-byte slot li of both operands
  const int ka = k0 + 4 * li, na = n0 + 4 * li;
  const bool a_ok = ka < K, y_ok = na < P.Nld;
  const float* a_col = P.A + (a_ok ? ka : 0);
  const float* y_col = P.Y + (y_ok ? na : 0);
  int n_m = m_begin;   // first row of the next stage to request
  auto issue_piece = [&](int g, int stage) {
    float* base = smem + stage * STAGE;
    const int i = g & 3;
    const int m = n_m + (wave * 4 + i) * 2 + kh;
    if (g < 4) {
      const int st = (int)((unsigned)m % (unsigned)T) + sh;
      const bool ok = a_ok && m < m_end && (unsigned)st < (unsigned)T;
      glds16(pick(a_col + (int64_t)(m + sh) * lda, ok), base + (wave * 4 + i) * 2 * 128);
    } else {
      const bool ok = y_ok && m < m_end;
      glds16(pick(y_col + (int64_t)m * ldy, ok), base + OP_FLOATS + (wave * 4 + i) * 2 * 128);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][jn][e] = 0.f;

  const uint32_t lds0 = lds_off(smem);
  const uint32_t a_rd = (uint32_t)(kh * 128 + 64 * wm + 2 * li) * 4u;
  const uint32_t b_rd = (uint32_t)(OP_FLOATS + kh * 128 + 64 * wn + 2 * li) * 4u;
  const bool do_bias = P.dbias != nullptr && bx == 0 && tap == 0;
  float bsum = 0.f;

  auto compute = [&](int stage, int fill_stage, auto fill_c) {
    constexpr bool fill = decltype(fill_c)::value;
    const uint32_t sb = lds0 + (uint32_t)stage * (STAGE * 4u);
    if (do_bias) {   // column sums of this stage's dY rows (masked rows are zeros); compiler-tracked reads, drained before the asm reads
      if (tid < 128) {
        const float* yc = smem + stage * STAGE + OP_FLOATS + tid;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int r = 0; r < BR; r += 2) {
          t0 += yc[r * 128];
          t1 += yc[(r + 1) * 128];
        }
        bsum += t0 + t1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    const uint32_t ab = sb + a_rd, bb = sb + b_rd;
    f32x2 a2[3], b2[3];
    a2[0] = dsr64<0>(ab);
    b2[0] = dsr64<0>(bb);
    a2[1] = dsr64<1024>(ab);
    b2[1] = dsr64<1024>(bb);
    static_for<0, 16>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      wait_lgkm<(s < 15 ? 2 : 0)>();   // reads of steps s + 1 (and s + 2's are not issued yet) may still be in flight
      __builtin_amdgcn_sched_barrier(0);
      const f32x2 av = a2[s % 3], bv = b2[s % 3];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);
      if (s + 2 < 16) {
        a2[(s + 2) % 3] = dsr64<(s + 2) * 1024>(ab);
        b2[(s + 2) % 3] = dsr64<(s + 2) * 1024>(bb);
      }
      if (s < NLD && fill) issue_piece(s, fill_stage);
    });
  };

#pragma unroll
  for (int g = 0; g < NLD; ++g) issue_piece(g, 0);
  n_m += BR;
  asm volatile("" ::"s"(T), "s"(K), "s"(lda), "s"(ldy), "s"(sh), "s"(m_end), "s"(nit), "s"(n_m));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int it = 0;
  for (; it + 1 < nit; ++it) {
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(it & 1, (it + 1) & 1, std::true_type{});
    n_m += BR;
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  compute(it & 1, 0, std::false_type{});

  // ---- epilogue: sub-tile (j, jn) element e -> row k0 + 64 wm + 2 (row_e) + j, column n0 + 64 wn + 2 li + jn ----
  float* W = P.W + (int64_t)tap * K * P.ldw;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + 64 * wm + 2 * ((e & 3) + 8 * (e >> 2) + 4 * kh) + j;
      if (k >= K) continue;
      const int n = n0 + 64 * wn + 2 * li;
      float* w = W + (int64_t)k * P.ldw + n;
      if (n < N) atomicAdd(w, acc[j][0][e]);
      if (n + 1 < N) atomicAdd(w + 1, acc[j][1][e]);
    }
  }
  if (do_bias && tid < 128 && n0 + tid < N) atomicAdd(P.dbias + n0 + tid, bsum);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct Variant {
  int bk, ns;
};
Variant env_variant() {   // read on every launch (tests and the tuning harness switch variants inside one process)
  Variant r{16, 4};
  if (const char* e = getenv("TACO_GEMM2_VARIANT")) {   // "<BK>x<stages>": 16x4 (default), 32x2 (forward conv banks), 32x3, 16x3, 16x5
    int bk = 0, ns = 0;
    if (sscanf(e, "%dx%d", &bk, &ns) == 2 && (bk == 16 || bk == 32) && ns >= 2 && ns <= 5) r = Variant{bk, ns};
  }
  return r;
}
template <int BK, int NS, int BX = 0>
int launch_variant(const Gemm2Args& g, int tiles, hipStream_t s) {
  constexpr size_t smem = (size_t)NS * ((size_t)TM * BK * sizeof(float) + (BX == 2 ? (size_t)kWImgTileBytes : (size_t)BK * TN * sizeof(float)));
  static DynSmemOnce once;
  TACO_REQUIRE(ensure_dyn_smem(once, reinterpret_cast<const void*>(conv_gemm2_kernel<BK, NS, BX>), smem),
               "conv_gemm2: cannot reserve %zu bytes of LDS", smem);
  TACO_KLAUNCH((conv_gemm2_kernel<BK, NS, BX>), dim3(tiles), dim3(256), smem, s, g);
  return TACO_OK;
}


// ---- weight images (kernels.h): builder kernel and the per-thread table ----
constexpr int kMaxWImgJobs = 40;
struct WImgJob {
  const float* W;
  unsigned short* img;
  int ldw, taps, K, N, kt, first;   // kt = k-tiles of 16 per tap (whole 32-deep units, as the GEMM kernel counts them); first = first block
};
struct WImgBatch {
  WImgJob j[kMaxWImgJobs];
  int n;
};
// one workgroup per (job, n-tile, k-tile): thread (j, li, kh) reads its 8 k-slots of column 128 nb + 4 li + j (coalesced: a wave reads
// 64 consecutive columns of one row per slot), splits them exactly as the GEMM kernel would (split8) and writes three 16-byte fragments
__global__ __launch_bounds__(256) void weight_image_kernel(WImgBatch B) {
  int ji = 0;
  for (int i = 1; i < B.n; ++i)
    if ((int)blockIdx.x >= B.j[i].first) ji = i;
  const WImgJob& J = B.j[ji];
  const int rel = blockIdx.x - J.first;
  const int its = J.taps * J.kt;
  const int nb = rel / its, it = rel - nb * its;
  const int tap = it / J.kt, k0 = (it - tap * J.kt) * 16;
  const int t = threadIdx.x, j = t & 3, li = (t >> 2) & 31, kh = t >> 7;
  const int n = nb * 128 + 4 * li + j;
  float w[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = k0 + 8 * (q >> 2) + 4 * kh + (q & 3);
    w[q] = (k < J.K && n < J.N) ? J.W[((int64_t)tap * J.K + k) * J.ldw + n] : 0.f;
  }
  const Pl3 p = split8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
  u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(J.img) + ((int64_t)nb * its + it) * kWImgTileBytes) + (j * 64 + kh * 32 + li);
  dst[0] = p.h;
  dst[4 * 64] = p.m;
  dst[8 * 64] = p.l;
}

struct WImgEntry {
  const float* W;
  const void* img;
  int ldw, taps, K, N, its;
};
constexpr int kMaxWImgEntries = 96;
thread_local WImgEntry g_wimg[kMaxWImgEntries];
thread_local int g_wimg_n = 0;
thread_local WImgBatch g_wimg_q;

}  // namespace

static inline int wimg_kt(int K) { return ((K + 31) / 32) * 2; }
int64_t weight_image_bytes(int taps, int K, int N) { return (int64_t)cdiv(N, 128) * taps * wimg_kt(K) * kWImgTileBytes; }
void weight_images_clear() {
  g_wimg_n = 0;
  g_wimg_q.n = 0;
}
int weight_image_add(const float* W, int ldw, int taps, int K, int N, void* img, bool queue_build, const float* src, int src_ldw) {
  TACO_REQUIRE(W && img && (reinterpret_cast<uintptr_t>(img) & 15) == 0 && taps > 0 && K > 0 && N > 0 && N <= ldw && (!src || N <= src_ldw),
               "weight_image_add: bad arguments");
  TACO_REQUIRE(g_wimg_n < kMaxWImgEntries, "weight_image_add: table full (%d)", kMaxWImgEntries);
  TACO_REQUIRE(weight_image_bytes(taps, K, N) < ((int64_t)1 << 31), "weight_image_add: image beyond 2 GiB");
  g_wimg[g_wimg_n++] = WImgEntry{W, img, ldw, taps, K, N, taps * wimg_kt(K)};
  if (queue_build) {
    TACO_REQUIRE(g_wimg_q.n < kMaxWImgJobs, "weight_image_add: build queue full (call weight_images_build every %d jobs)", kMaxWImgJobs);
    WImgJob& j = g_wimg_q.j[g_wimg_q.n++];
    j.W = src ? src : W; j.img = reinterpret_cast<unsigned short*>(img); j.ldw = src ? src_ldw : ldw; j.taps = taps; j.K = K; j.N = N; j.kt = wimg_kt(K); j.first = 0;
  }
  return TACO_OK;
}
int weight_images_build(hipStream_t s) {
  if (g_wimg_q.n == 0) return TACO_OK;
  int blocks = 0;
  for (int i = 0; i < g_wimg_q.n; ++i) {
    WImgJob& j = g_wimg_q.j[i];
    j.first = blocks;
    blocks += cdiv(j.N, 128) * j.taps * j.kt;
  }
  TACO_KLAUNCH(weight_image_kernel, dim3(blocks), dim3(256), 0, s, g_wimg_q);
  TACO_LAUNCH_CHECK("weight_image_kernel");
  g_wimg_q.n = 0;
  return TACO_OK;
}
const void* weight_image_find(const float* W, int ldw, int taps, int K, int N, int* its) {
  for (int i = 0; i < g_wimg_n; ++i) {
    const WImgEntry& e = g_wimg[i];
    // (columns between the registered N and the end of its last 4-column group are zeros of the image: a launch whose loadable
    //  width is N rounded up to the float4 contract -- the final dense layer: 1025 -> 1028 -- reads the same values it would
    //  read from its zero-padded fp32 copy)
    if (e.W == W && e.ldw == ldw && e.taps == taps && e.K == K && N <= (e.N + 3) / 4 * 4) {
      *its = e.its;
      return e.img;
    }
  }
  return nullptr;
}

int gemm2_min_tiles() {
  const char* e = getenv("TACO_GEMM2_MIN_TILES");   // 0 disables the second-generation kernel
  return e ? atoi(e) : 160;
}

// Returns TACO_ENOTFOUND (nothing launched) when the batch does not meet the DMA contract or is too small to fill the chip
// with 128 x 128 tiles; the caller then falls back to conv_gemm_kernel.
// debug: only the eligible launches whose running index falls in [lo, hi) use the new kernel (bisecting a divergence)
static int g_win_lo = 0, g_win_hi = 1 << 30, g_win_idx = 0;
static int64_t g_img_launches = 0;   // launches that took the B-image form since the last taco_debug_weight_image(NULL, ...)
extern "C" __attribute__((visibility("default"))) int taco_debug_gemm2_window(int lo, int hi) {
  g_win_lo = lo; g_win_hi = hi;
  const int n = g_win_idx;
  g_win_idx = 0;
  return n;   // eligible launches seen since the last call
}

// Op-level access to the weight images (tests, tools): W == null clears this thread's table; img == null returns the bytes
// image(W) needs; otherwise registers image(W) at img and builds it on `stream` -- the next taco_conv_gemm calls of this thread
// whose weights are W then run the B-image form of the kernel.
extern "C" __attribute__((visibility("default"))) int64_t taco_debug_weight_image(const float* W, int ldw, int taps, int K, int N, void* img,
                                                                                  int64_t img_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!W) {
    weight_images_clear();
    const int64_t n = g_img_launches;
    g_img_launches = 0;
    return n;
  }
  if (taps <= 0 || K <= 0 || N <= 0) return TACO_EINVAL;
  const int64_t need = weight_image_bytes(taps, K, N);
  if (!img) return need;
  if (img_bytes < need) return TACO_EINVAL;
  const int rc = weight_image_add(W, ldw, taps, K, N, img, true);
  if (rc != TACO_OK) return rc;
  return weight_images_build(stream);
}

// m-tiles of a problem: pooled problems advance by TM - 1 rows (the last row of a tile is the next tile's first)
static inline int m_tiles(const ConvGemmProblem& p) { return p.pool ? cdiv(p.M > 1 ? p.M - 1 : 1, TM - 1) : cdiv(p.M, TM); }
static bool pool_contract(const ConvGemmProblem& p) {
  const bool common = p.N % 4 == 0 && p.ldc % 4 == 0 && al16(p.C) && (!p.Cpre || al16(p.Cpre)) && !p.keep && !p.residual &&
                      p.bias_stride == 0 && !p.atomic_out && p.it0 == 0 && p.it1 == 0;
  if (p.pool == 2)
    return common && p.pool_x && al16(p.pool_x) && p.pool_dgamma && p.pool_dbeta && p.scale && p.shift && !p.Cpre && !p.bias &&
           p.act == TACO_ACT_NONE;
  return common;
}
// the kernel addresses both operands with 32-bit byte offsets against one buffer descriptor each (range 2 GiB)
static bool dma_contract(const ConvGemmProblem& p) {
  const int64_t lim = (int64_t)1 << 31;
  return ((int64_t)p.M + p.taps + 1) * p.lda * 4 + (int64_t)p.K * 4 < lim && ((int64_t)p.taps * p.K + 32) * p.ldw * 4 < lim &&
         (p.bank_filters == 0 || (p.taps == p.bank_filters * (p.bank_filters + 1) / 2 && p.K % 32 == 0 &&
                                  p.pad_l == (p.bank_filters - 1) - (p.bank_filters - 1) / 2 && p.lda >= p.bank_filters * p.K));
}
bool conv_gemm2_would_launch(const ConvGemmBatch& batch) {
  const int min_tiles = gemm2_min_tiles();
  if (min_tiles <= 0) return false;
  int tiles = 0;
  for (int i = 0; i < batch.n; ++i) {
    const ConvGemmProblem& p = batch.p[i];
    if ((p.flags & 3) != 3 || !dma_contract(p) || (p.pool && !pool_contract(p))) return false;
    tiles += m_tiles(p) * cdiv(p.N, TN);
  }
  return tiles >= min_tiles;
}

int launch_conv_gemm2(ConvGemmBatch& batch, hipStream_t stream, bool force) {
  const int min_tiles = force ? 1 : gemm2_min_tiles();
  if (gemm2_min_tiles() <= 0) return TACO_ENOTFOUND;
  Gemm2Args g;
  int order[kMaxGemmBatch];
  int tiles = 0;
  bool any_pool = false;
  for (int i = 0; i < batch.n; ++i) {
    ConvGemmProblem& p = batch.p[i];
    if ((p.flags & 3) != 3 || !dma_contract(p)) return TACO_ENOTFOUND;   // both operands: 16-byte aligned rows, K / Nld multiples of 4, < 2 GiB
    if (p.pool) {
      TACO_REQUIRE(pool_contract(p), "conv_gemm2: pooled epilogue needs float4 rows and no keep / residual / row bias / atomics / k-split");
      any_pool = true;
    }
    tiles += m_tiles(p) * cdiv(p.N, TN);
    order[i] = i;
  }
  if (tiles < min_tiles) return TACO_ENOTFOUND;
  if (!any_pool) {   // (pooled batches were promised to their caller by conv_gemm2_would_launch: no debug window for them)
    const int idx = g_win_idx++;
    if (getenv("TACO_GEMM2_TRACE"))
      for (int i = 0; i < batch.n; ++i)
        fprintf(stderr, "gemm2 #%d.%d M=%d N=%d K=%d taps=%d T=%d pad_l=%d act=%d keep=%d res=%d pre=%d aff=%d atomic=%d bias_stride=%d lda=%d ldw=%d ldc=%d\n",
                idx, i, batch.p[i].M, batch.p[i].N, batch.p[i].K, batch.p[i].taps, batch.p[i].T, batch.p[i].pad_l, batch.p[i].act,
                batch.p[i].keep != nullptr, batch.p[i].residual != nullptr, batch.p[i].Cpre != nullptr,
                batch.p[i].scale != nullptr, batch.p[i].atomic_out, batch.p[i].bias_stride, batch.p[i].lda, batch.p[i].ldw, batch.p[i].ldc);
    if (idx < g_win_lo || idx >= g_win_hi) return TACO_ENOTFOUND;
  }
  // longest k-loops first: the hardware hands tiles to workgroup slots in grid order, so the short problems fill the tail
  std::stable_sort(order, order + batch.n, [&](int a, int b) {
    return (int64_t)batch.p[a].taps * batch.p[a].K > (int64_t)batch.p[b].taps * batch.p[b].K;
  });
  g.batch.n = batch.n;
  int first = 0;
  for (int i = 0; i < batch.n; ++i) {
    ConvGemmProblem p = batch.p[order[i]];
    // bit 2: float4 epilogue (every row of C / Cpre / residual 16-byte aligned, keep mask readable as one dword per 4 columns)
    const bool vec = p.N % 4 == 0 && p.ldc % 4 == 0 && al16(p.C) && (!p.Cpre || al16(p.Cpre)) &&
                     (!p.residual || (p.ldr % 4 == 0 && al16(p.residual))) &&
                     (!p.keep || (reinterpret_cast<uintptr_t>(p.keep) & 3) == 0);
    // bit 3: shifted float4 epilogue for row pitches that are not multiples of 4 (plain bias / activation / affine outputs only)
    const bool shifted = !vec && !p.pool && al16(p.C) && p.ldc % 4 != 0 && !p.Cpre && !p.residual && !p.keep && !p.atomic_out &&
                         p.bias_stride == 0 && p.it0 == 0 && p.it1 == 0;
    p.flags = (p.flags & 3) | (vec ? 4 : 0) | (shifted ? 8 : 0);
    g.batch.p[i] = p;
    g.first[i] = first;
    g.mt[i] = m_tiles(p);
    first += g.mt[i] * cdiv(p.N, TN);
  }
  // XCD-aware order when every tile of the grid runs the same number of k-tiles (TACO_GEMM2_XCD=0: the old order, A/B runs)
  {
    bool same = true;
    auto depth = [](const ConvGemmProblem& p) {
      return p.it1 > 0 ? (int64_t)(p.it1 - p.it0) : (int64_t)p.taps * ((p.K + 31) / 32);
    };
    for (int i = 1; i < batch.n; ++i) same = same && depth(g.batch.p[i]) == depth(g.batch.p[0]);
    const char* e = getenv("TACO_GEMM2_XCD");
    g.xcd_map = (same && !(e && atoi(e) == 0)) ? 1 : 0;
    // conv-bank order: 16 or 8 problems of different depth with the same number of tiles each (g.batch.p is sorted deepest first)
    const char* eb = getenv("TACO_GEMM2_BANK_XCD");
    bool bank = !same && !(eb && atoi(eb) == 0) && (batch.n == 16 || batch.n == 8);
    const int t = g.mt[0] * cdiv(g.batch.p[0].N, TN);
    for (int i = 1; i < batch.n && bank; ++i) bank = g.mt[i] * cdiv(g.batch.p[i].N, TN) == t;
    if (bank) {
      g.xcd_map = 2;
      const int h = (t + 1) / 2;
      for (int x = 0; x < 8; ++x) {
        if (batch.n == 16) {
          g.bk_pa[x] = x; g.bk_a0[x] = 0; g.bk_na[x] = t;
          g.bk_pb[x] = 15 - x; g.bk_b0[x] = 0; g.bk_nb[x] = t;
        } else {
          const int pr = x >> 1, lo = (x & 1) ? h : 0, cnt = (x & 1) ? t - h : h;
          g.bk_pa[x] = pr; g.bk_a0[x] = lo; g.bk_na[x] = cnt;
          g.bk_pb[x] = 7 - pr; g.bk_b0[x] = lo; g.bk_nb[x] = cnt;
        }
      }
      tiles = batch.n == 16 ? 16 * t : 8 * 2 * h;   // grid: 8 XCDs x the longest share
    }
  }
  Variant v = env_variant();
  if (!getenv("TACO_GEMM2_VARIANT")) {
    // Both forms hold 64 KB of LDS (two workgroups per CU).  Four 16-deep stages keep 48 k-columns in flight instead of 32:
    // inside a train step the operands come from HBM / the Infinity Cache, not from a warm L2, and the deeper ring is what
    // the launches with long rows wait less with (same-box family traces, profiles/r04_gemm_variants.txt: post-net proj1
    // input gradient 221 -> 186 us, proj1 k-split 210-230 -> 202; everything else within 2 %).  K = 80 (post-net conv bank) is 2.5
    // tiles of 32 and needs the 16-deep form anyway.  The forward conv bank of the encoder (K = 128 per tap, L2-resident
    // working set) stays on the 32-deep form: 258 vs 265 us.
    bool bank32 = true;
    for (int i = 0; i < batch.n; ++i) bank32 = bank32 && batch.p[i].pool == 1 && batch.p[i].K % 32 == 0;
    v = bank32 ? Variant{32, 2} : Variant{16, 4};
  }
  // chain length of the deepest problem in k-products (k-split chunks count with their own [it0, it1) range of 32-deep k-tiles)
  int64_t chain = 0;
  for (int i = 0; i < g.batch.n; ++i) {
    const ConvGemmProblem& p = g.batch.p[i];
    const int64_t c = p.it1 > 0 ? (int64_t)(p.it1 - p.it0) * 32 : (int64_t)p.taps * p.K;
    chain = c > chain ? c : chain;
  }
  if (env_bf16x() && chain <= bf16x_max_chain()) {   // 16-deep tiles only (K = 80, K % 32 != 0 and the forward banks alike); the ring depth follows the variant
    // every problem's weights have a pre-split plane image (weight_image_add): the B-image form.  TACO_GEMM2_BSPLIT=0: never (A/B runs).
    const char* eb2 = getenv("TACO_GEMM2_BSPLIT");
    bool img = !(eb2 && atoi(eb2) == 0);
    for (int i = 0; i < g.batch.n && img; ++i) {
      ConvGemmProblem& p = g.batch.p[i];
      p.Wimg = weight_image_find(p.W, p.ldw, p.taps, p.K, p.Nld > 0 ? p.Nld : p.N, &p.img_its);
      img = p.Wimg != nullptr;
    }
    if (img) {
      __atomic_fetch_add(&g_img_launches, (int64_t)1, __ATOMIC_RELAXED);
      // 20 KB per stage: three stages = 60 KB (two workgroups per CU), four = 80 KB (2 x 80 = the CU's whole 160 KB)
      static const int bi_ns = [] { const char* e = getenv("TACO_GEMM2_BI_NS"); return e ? atoi(e) : 3; }();
      if (bi_ns == 4) return launch_variant<16, 4, 2>(g, tiles, stream);
      return launch_variant<16, 3, 2>(g, tiles, stream);
    }
    if (v.bk == 16 && v.ns == 3) return launch_variant<16, 3, 1>(g, tiles, stream);
    if (v.bk == 16 && v.ns == 5) return launch_variant<16, 5, 1>(g, tiles, stream);
    return launch_variant<16, 4, 1>(g, tiles, stream);
  }
  if (v.bk == 32 && v.ns == 2) return launch_variant<32, 2>(g, tiles, stream);
  if (v.bk == 32 && v.ns == 3) return launch_variant<32, 3>(g, tiles, stream);
  if (v.bk == 16 && v.ns == 3) return launch_variant<16, 3>(g, tiles, stream);
  if (v.bk == 16 && v.ns == 4) return launch_variant<16, 4>(g, tiles, stream);
  if (v.bk == 16 && v.ns == 5) return launch_variant<16, 5>(g, tiles, stream);
  return launch_variant<32, 2>(g, tiles, stream);
}

// Grouped weight-gradient launch on gemm_tn2_kernel.  `probs` must already carry flags / splits / chunk (plan below).
// Returns the number of problems it took (they are removed from the caller's list by index order), or a negative error.
static void plan_tn2(GemmTnArgs& a, int& gx, int& gy, int& blocks) {
  gx = cdiv(a.K, 128);
  gy = cdiv(a.N, 128);
  const int64_t tiles = (int64_t)gx * gy * a.taps;
  int splits = (int)((640 + tiles - 1) / tiles);   // ~2.5 workgroups per CU over the whole launch group
  const int max_splits = cdiv(a.M, 256);           // at least 8 stages of 32 rows per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int chunk = cdiv(a.M, splits);
  chunk = cdiv(chunk, 32) * 32;
  splits = cdiv(a.M, chunk);
  a.splits = splits;
  a.chunk = chunk;
  blocks = (int)tiles * splits;
}

bool gemm_tn2_eligible(const GemmTnArgs& a) {
  // OPT-IN (TACO_TN2=1): measured on MI355X this kernel is correct but SLOWER than gemm.hip's gemm_tn at every model shape
  // (e.g. post proj1 dW 207 vs 199 us, dense dW 140 vs 71 us, bank dW 70 vs 43 us): the weight gradients are split-M launches
  // whose workgroups run only 8-12 stages before a 128 x 128 atomic epilogue, and the interleaved column map of the b64
  // fragment reads turns that epilogue into two half-used-cache-line atomics per row.  Kept for the tuning harness.
  const char* e2 = getenv("TACO_TN2");   // read on every call (the A/B harness toggles it inside one process)
  const bool on = e2 && atoi(e2) != 0;
  if (!on || gemm2_min_tiles() <= 0 || taco_deterministic()) return false;
  if (a.batch != 1 || a.K < 96 || a.N < 96 || a.M < 512) return false;
  if (a.lda % 4 || a.ldy % 4 || a.K % 4 || !al16(a.A) || !al16(a.Y)) return false;
  const int nld = a.Nld > 0 ? a.Nld : (a.N % 4 == 0 ? a.N : 0);
  return nld > 0 && nld % 4 == 0 && nld <= a.ldy;
}

int launch_gemm_tn2(const GemmTnArgs* probs, int n, hipStream_t stream) {
  TACO_REQUIRE(n >= 1 && n <= kMaxTnBatch, "gemm_tn2: %d problems out of range", n);
  Tn2Args g;
  g.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    GemmTnArgs a = probs[i];
    if (a.Nld <= 0) a.Nld = a.N;
    int nb = 0;
    plan_tn2(a, g.gx[i], g.gy[i], nb);
    g.p[i] = a;
    g.first[i] = blocks;
    blocks += nb;
  }
  TACO_KLAUNCH(gemm_tn2_kernel, dim3(blocks), dim3(256), 2 * 2 * 32 * 128 * sizeof(float), stream, g);
  return TACO_OK;
}
