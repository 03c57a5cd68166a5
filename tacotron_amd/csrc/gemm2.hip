// gemm2.hip -- second-generation NN conv-GEMM for the big feed-forward launches (same contract as conv_gemm_kernel in gemm.hip:
//   C = post(act(sum_tap shift_tap(A) . W[tap] + bias)), tf.layers.dense / conv1d 'same', ops.py:54-60,80-86, tacotron.py:40-43,148).
//
// What changed against gemm.hip (47 % matrix-pipe utilisation, one barrier per 16-deep k-tile, register-staged
// loads, transposing ds_write_b32 stores):
//   * both operands travel HBM/L2 -> LDS by direct DMA (global_load_lds_dwordx4, 1 KiB per wave instruction): no staging
//     registers, no ds_write pass, no VALU selects.  Masked elements (rows outside the sequence for a shifted tap, the K / N /
//     M tails) are redirected PER LANE to a 16-byte zero word, so the DMA image is always complete.
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

#include "common.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) float g_zero4[4];   // zero-initialised device word: source of every masked DMA lane

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int TM = 128, TN = 128;

struct Gemm2Args {
  ConvGemmBatch batch;
  int first[kMaxGemmBatch];   // first linear tile of each problem
  int mt[kMaxGemmBatch];      // m-tiles of each problem (m runs fastest inside a problem)
};

template <int BK, int NS>
__global__ __launch_bounds__(256, 2) void conv_gemm2_kernel(Gemm2Args G) {
  constexpr int SLOTS = BK / 4;               // 16-byte slots per A row
  constexpr int A_RPI = 64 / SLOTS;           // A rows per wave instruction: 8 (BK = 32) / 16 (BK = 16)
  constexpr int A_INSTR = TM / A_RPI / 4;     // per wave
  constexpr int B_INSTR = BK / 2 / 4;         // one instruction = 2 k-rows of 128 floats; per wave
  constexpr int NLD = A_INSTR + B_INSTR;      // DMA instructions per wave per k-tile
  constexpr int A_FLOATS = TM * BK, B_FLOATS = BK * TN, STAGE = A_FLOATS + B_FLOATS;
  constexpr int SWZ_SH = BK == 32 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // NS * STAGE floats, the ONLY LDS object of the kernel

  int pi = 0;
  for (int i = 1; i < G.batch.n; ++i)
    if ((int)blockIdx.x >= G.first[i]) pi = i;
  const ConvGemmProblem& P = G.batch.p[pi];
  const int rel = blockIdx.x - G.first[pi];
  const int mtiles = G.mt[pi];
  const int tnn = rel / mtiles, tmm = rel - tnn * mtiles;
  const int m0 = tmm * TM, n0 = tnn * TN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = P.T, K = P.K, lda = P.lda, ldw = P.ldw;

  // ---- DMA source setup: each thread serves the same A rows / B slots for every k-tile ----
  const float* a_ptr[A_INSTR];
  int a_t[A_INSTR], a_k[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * A_RPI + lane / SLOTS;
    const int q = (lane % SLOTS) ^ ((row >> SWZ_SH) & (SLOTS - 1));   // which logical slot lands in this lane's LDS slot
    const int m = m0 + row;
    const bool ok = m < P.M;
    a_t[i] = ok ? m % T : -(1 << 28);
    a_k[i] = 4 * q;
    a_ptr[i] = P.A + (int64_t)(ok ? m : 0) * lda + 4 * q;
  }
  const int b_c = n0 + 4 * (lane & 31);
  const bool b_ok = b_c < P.Nld;
  const int b_r0 = wave * B_INSTR * 2 + (lane >> 5);
  const float* b_ptr = P.W + (int64_t)b_r0 * ldw + (b_ok ? b_c : 0);

  const int ktiles = (K + BK - 1) / BK;
  const int nit = P.taps * ktiles;
  const int pad_l = P.pad_l;
  // the zero word's address, kept in a VGPR pair (opaque to the optimiser: otherwise it is re-fetched from the GOT with an
  // s_load + s_waitcnt lgkmcnt(0) in front of every DMA instruction, and lgkmcnt also counts the LDS fragment reads)
  const float* zero = g_zero4;
  asm volatile("" : "+v"(zero));

  // next tile to issue: (tap, k0) advance incrementally (no division on the loop path)
  int n_tap = 0, n_k0 = 0;
  auto issue = [&](int stage) {
    const int sh = n_tap - pad_l;
    const int k0 = n_k0;
    float* As = smem + stage * STAGE;
    float* Bs = As + A_FLOATS;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const bool ok = (unsigned)(a_t[i] + sh) < (unsigned)T && k0 + a_k[i] < K;
      const float* src = ok ? a_ptr[i] + ((int64_t)sh * lda + k0) : zero;
      glds16(src, As + (wave * A_INSTR + i) * A_RPI * BK);
    }
    const float* wt = b_ptr + ((int64_t)n_tap * K + k0) * ldw;
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
      const bool ok = b_ok && k0 + b_r0 + 2 * i < K;
      const float* src = ok ? wt + (int64_t)(2 * i) * ldw : zero;
      glds16(src, Bs + (wave * B_INSTR + i) * 2 * TN);
    }
    n_k0 += BK;
    if (n_k0 >= K) {
      n_k0 = 0;
      ++n_tap;
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int li = lane & 31, kh = lane >> 5;
  const int arow = wave * 32 + li;
  const int aswz = (arow >> SWZ_SH) & (SLOTS - 1);

  // 64 MFMAs per 32-deep tile; the LDS fragments are requested TWO (p, c) steps ahead of the MFMAs that consume them, so a
  // read has four 64-cycle MFMAs to land and the wait in front of a step is a counted lgkmcnt, not lgkmcnt(0)
  auto compute = [&](int stage) {
    const float* As = smem + stage * STAGE + arow * BK;
    const float* Bs = smem + stage * STAGE + A_FLOATS + 4 * li;
    constexpr int NP = BK / 8;
    f32x4 a4[2], b4[3];
    auto lda4 = [&](int p) { return *reinterpret_cast<const f32x4*>(As + 4 * ((2 * p + kh) ^ aswz)); };
    auto ldb4 = [&](int idx) { return *reinterpret_cast<const f32x4*>(Bs + (8 * (idx >> 2) + 4 * kh + (idx & 3)) * TN); };
    a4[0] = lda4(0);
    b4[0] = ldb4(0);
    b4[1] = ldb4(1);
#pragma unroll
    for (int idx = 0; idx < NP * 4; ++idx) {
      const int p = idx >> 2, c = idx & 3;
      // fragments of step idx + 2 (B) / of the next p (A, two steps before its first use) are requested now
      if (idx + 2 < NP * 4) b4[(idx + 2) % 3] = ldb4(idx + 2);
      if (c == 2 && p + 1 < NP) a4[(p + 1) & 1] = lda4(p + 1);
      // (pinned: the machine scheduler otherwise sinks each read back in front of its first use and waits lgkmcnt(0) there)
      __builtin_amdgcn_sched_barrier(0);
      const float av = a4[p & 1][c];
      const f32x4 bv = b4[idx % 3];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[3], acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- NS-stage ring: tile t lives in stage t % NS; tiles up to t + NS - 2 are in flight while t is computed ----
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nit) issue(s);
  for (int it = 0; it < nit; ++it) {
    // this wave's share of tile `it` has landed once at most the younger tiles' DMA instructions are outstanding
    if (nit - 1 - it >= NS - 2) wait_vm<NLD*(NS - 2)>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every wave's share has landed AND every wave has finished reading tile it - 1
    asm volatile("" ::: "memory");
    if (it + NS - 1 < nit) issue((it + NS - 1) % NS);   // refills the stage tile it - 1 just vacated
    compute(it % NS);
  }

  // ---- epilogue.  C/D layout of v_mfma_f32_32x32x2: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5);
  //      sub-tile j holds columns n0 + 4 li + j, so element e of the four accumulators is one float4 of row `row` ----
  const int n = n0 + 4 * li;
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

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct Variant {
  int bk, ns;
};
Variant env_variant() {   // read on every launch (tests and the tuning harness switch variants inside one process)
  Variant r{32, 2};
  if (const char* e = getenv("TACO_GEMM2_VARIANT")) {   // "<BK>x<stages>": 32x2 (default), 32x3, 16x3, 16x4
    int bk = 0, ns = 0;
    if (sscanf(e, "%dx%d", &bk, &ns) == 2 && (bk == 16 || bk == 32) && ns >= 2 && ns <= 4) r = Variant{bk, ns};
  }
  return r;
}

template <int BK, int NS>
int launch_variant(const Gemm2Args& g, int tiles, hipStream_t s) {
  constexpr size_t smem = (size_t)NS * (TM * BK + BK * TN) * sizeof(float);
  static const bool ok = smem <= 64 * 1024 ||
                         hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm2_kernel<BK, NS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
  TACO_REQUIRE(ok, "conv_gemm2: cannot reserve %zu bytes of LDS", smem);
  hipLaunchKernelGGL((conv_gemm2_kernel<BK, NS>), dim3(tiles), dim3(256), smem, s, g);
  return TACO_OK;
}

}  // namespace

int gemm2_min_tiles() {
  const char* e = getenv("TACO_GEMM2_MIN_TILES");   // 0 disables the second-generation kernel
  return e ? atoi(e) : 96;
}

// Returns TACO_ENOTFOUND (nothing launched) when the batch does not meet the DMA contract or is too small to fill the chip
// with 128 x 128 tiles; the caller then falls back to conv_gemm_kernel.
// debug: only the eligible launches whose running index falls in [lo, hi) use the new kernel (bisecting a divergence)
static int g_win_lo = 0, g_win_hi = 1 << 30, g_win_idx = 0;
extern "C" __attribute__((visibility("default"))) int taco_debug_gemm2_window(int lo, int hi) {
  g_win_lo = lo; g_win_hi = hi;
  const int n = g_win_idx;
  g_win_idx = 0;
  return n;   // eligible launches seen since the last call
}

int launch_conv_gemm2(ConvGemmBatch& batch, hipStream_t stream) {
  const int min_tiles = gemm2_min_tiles();
  if (min_tiles <= 0) return TACO_ENOTFOUND;
  Gemm2Args g;
  int order[kMaxGemmBatch];
  int tiles = 0;
  for (int i = 0; i < batch.n; ++i) {
    ConvGemmProblem& p = batch.p[i];
    if ((p.flags & 3) != 3) return TACO_ENOTFOUND;   // both operands: 16-byte aligned rows, K / Nld multiples of 4
    tiles += cdiv(p.M, TM) * cdiv(p.N, TN);
    order[i] = i;
  }
  if (tiles < min_tiles) return TACO_ENOTFOUND;
  {
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
    p.flags = (p.flags & 3) | (vec ? 4 : 0);
    g.batch.p[i] = p;
    g.first[i] = first;
    g.mt[i] = cdiv(p.M, TM);
    first += g.mt[i] * cdiv(p.N, TN);
  }
  const Variant v = env_variant();
  if (v.bk == 32 && v.ns == 2) return launch_variant<32, 2>(g, tiles, stream);
  if (v.bk == 32 && v.ns == 3) return launch_variant<32, 3>(g, tiles, stream);
  if (v.bk == 16 && v.ns == 3) return launch_variant<16, 3>(g, tiles, stream);
  if (v.bk == 16 && v.ns == 4) return launch_variant<16, 4>(g, tiles, stream);
  return launch_variant<32, 2>(g, tiles, stream);
}
