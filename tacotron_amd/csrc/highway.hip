// highway.hip -- the CBHG's four highway layers (ops.py:27-46, 97-107) as ONE kernel per direction of the pass.
//
// The layers are row-wise: y = relu(x Wh + bh) * sigmoid(x Wt + bt) + x * (1 - sigmoid(x Wt + bt)), y feeds the next layer.
// Run as separate GEMMs each layer is a 0.4-0.75 GFLOP launch with K = 128 that cannot fill the chip (16-26 TF measured)
// plus an elementwise blend; here a workgroup keeps a 32-row tile resident in LDS and walks all layers.  MFMA tiling: 4 waves,
// wave w owns the 32 gate columns [32w, 32w+32) AND the same 32 candidate columns, so T and H of an element meet in one lane
// and the blend is the epilogue.  The backward kernel does the same for the activation-gradient chain; the weight gradients
// stay grouped TN GEMMs over the stashed d[T|H] (model.hip).
//
// Round 3: the weights no longer pass through LDS.  The B operand of v_mfma_f32_32x32x2_f32 is one float per lane --
// W[k = 2j + (lane >> 5)][n = 32 wave + (lane & 31)] -- i.e. a coalesced global_load_dword straight into the register the MFMA
// reads; every workgroup reads the same 128 KB per layer, so these are L2 hits.  They are fetched in chunks of 32 k-values
// (16 or 32 registers per operand), one chunk ahead of the MFMAs that use them, across layer boundaries too: no W staging, no
// barrier inside a layer's k-loop (round 2: one per 16 MFMAs, three ds_read_b32 in front of every MFMA pair, 12-15 % of the
// matrix pipe).  Only the activations live in LDS (feature-major, pitch 33); two barriers per layer guard their update.
// Barriers are LDS-only (`s_waitcnt lgkmcnt(0); s_barrier`): a __syncthreads() would drain the weight prefetch.
// Measured around it (same box): the stash stores of [T | H] and y are 10 of the forward kernel's 39 us at M = 6400 (39 MB per
// launch; issuing them in three batches behind the next layer's prefetches changed nothing -- it is write bandwidth, not the
// in-order counter); reading all A fragments of a layer in one burst changed nothing either.
// Accumulators are split in two independent chains so that back-to-back MFMAs never wait for each other's result.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void hw_static_for(F&& f) {   // compile-time chunk index: register sets, layers and stages fold per chunk
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    hw_static_for<I + 1, N>(f);
  }
}

constexpr int HB = 32;         // rows per workgroup
constexpr int HC = 128;        // highway width
constexpr int HPAD = HB + 1;   // activation row pitch, feature-major (conflict-free column reads)
constexpr int FCK = 16;        // forward: k-pairs per chunk (32 k-values): 4 chunks per layer
constexpr int BCK = 32;        // backward: k-pairs per chunk (64 of the 256 k-values): 4 chunks per layer

// forward: grid = ceil(M / 32), block = 256.
// AD (round 6; multi-speaker encoder, ops.py:97-107): every layer has an input adapter in front of its gates,
//   x_l = h_l . Wa_l[:128] + rowb_l[sequence]      (rowb = relu(dense(speaker)) . Wa_l[128:] + ba_l, a per-sequence bias: model.hip),
// run as one more 128 x 128 GEMM stage per layer inside this launch (its output tile goes to a second LDS buffer and, as the
// backward pass's stash, to hx[l]); rounds 1-5 ran the speaker model's layers as separate adapter / gate / blend launches.
template <bool AD>
__global__ __launch_bounds__(256, 2) void highway_stack_fwd_kernel(HighwayStackArgs a) {
  __shared__ __attribute__((aligned(16))) float hT[HC][HPAD];          // current layer input, feature-major
  __shared__ __attribute__((aligned(16))) float xT[AD ? HC : 1][HPAD];  // AD: the adapter's output = the gates' input
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * HB;
  const int n = 32 * wave + li;

  // input tile -> hT (transposed)
  {
    const int row = tid >> 3, kq = tid & 7;
    const int m = m0 + row;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (kq + 8 * i) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M) v = *reinterpret_cast<const float4*>(a.x + (int64_t)m * HC + k);
      hT[k + 0][row] = v.x;
      hT[k + 1][row] = v.y;
      hT[k + 2][row] = v.z;
      hT[k + 3][row] = v.w;
    }
  }
  // B fragments of a chunk: this lane's [Wt | Wh] (or, adapter chunks, Wa) values for k = 2 (16 c + j) + lk, column n
  constexpr int CPL = AD ? 8 : 4;   // chunks per layer: (AD: 4 adapter chunks, then) 4 gate / candidate chunks
  float wt_r[2][FCK], wh_r[2][FCK];
  auto load_chunk = [&](float (&t)[FCK], float (&h)[FCK], int l, int cc) {
    const int c = cc & 3;
    if (AD && cc < 4) {
      const float* wa = a.wa[l] + (2 * c * FCK + lk) * HC + n;
#pragma unroll
      for (int j = 0; j < FCK; ++j) t[j] = wa[2 * j * HC];
    } else {
      const float* wt = a.wt[l] + (2 * c * FCK + lk) * HC + n;
      const float* wh = a.wh[l] + (2 * c * FCK + lk) * HC + n;
#pragma unroll
      for (int j = 0; j < FCK; ++j) {
        t[j] = wt[2 * j * HC];
        h[j] = wh[2 * j * HC];
      }
    }
  };
  load_chunk(wt_r[0], wh_r[0], 0, 0);

  f32x16 accT[2], accH[2];
#pragma unroll
  for (int i = 0; i < 4 * CPL; ++i) {   // chunk i = (layer i / CPL, chunk i % CPL); fully unrolled: register sets by parity
    const int l = i / CPL, cc = i % CPL, c = cc & 3;
    const bool ad = AD && cc < 4;
    if (l >= a.nl) break;
    if (i + 1 < 4 * CPL && (i + 1) / CPL < a.nl) load_chunk(wt_r[(i + 1) & 1], wh_r[(i + 1) & 1], (i + 1) / CPL, (i + 1) % CPL);
    __builtin_amdgcn_sched_barrier(0);   // the prefetch stays HERE: the scheduler would sink each load to just in front of its MFMA
    if (c == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) accT[0][e] = accT[1][e] = accH[0][e] = accH[1][e] = 0.f;
      lds_barrier();   // the stage's input tile is complete (first layer: the input tile; later: the previous stage's output)
    }
    float av[FCK];
#pragma unroll
    for (int j = 0; j < FCK; ++j) av[j] = (AD && !ad) ? xT[2 * (c * FCK + j) + lk][li] : hT[2 * (c * FCK + j) + lk][li];
#pragma unroll
    for (int j = 0; j < FCK; ++j) {
      accT[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wt_r[i & 1][j], accT[j & 1], 0, 0, 0);
      if (!ad) accH[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wh_r[i & 1][j], accH[j & 1], 0, 0, 0);
    }
    if (c == 3 && ad) {
      // adapter epilogue: x = h . Wa[:128] + rowb[sequence of the row]; stash (hx) + the gates' input tile.  No barrier in front of the
      // xT writes: its last readers (the previous layer's gate stage) all passed that stage's epilogue barrier.
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int m = m0 + r;
        const int mm = m < a.M ? m : a.M - 1;
        const float x = accT[0][e] + accT[1][e] + a.rowb[l][(int64_t)(mm / a.T) * HC + n];
        if (m < a.M) a.hx[l][(int64_t)m * HC + n] = x;
        xT[n][r] = x;
      }
    }
    if (c == 3 && !ad) {
      // epilogue: C layout col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
      const float bT = a.bt[l][n], bH = a.bh[l][n];
      float y[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int m = m0 + r;
        const float T = sigmoid_fast(accT[0][e] + accT[1][e] + bT);   // v_exp + v_rcp, as in the decoder kernels
        const float H = fmaxf(accH[0][e] + accH[1][e] + bH, 0.f);
        const float h = AD ? xT[n][r] : hT[n][r];
        y[e] = H * T + h * (1.f - T);
        if (m < a.M) {
          if (a.th[l]) {   // gate / candidate stash for the backward pass (null at inference)
            a.th[l][(int64_t)m * 2 * HC + n] = T;
            a.th[l][(int64_t)m * 2 * HC + HC + n] = H;
          }
          a.y[l][(int64_t)m * HC + n] = y[e];
        }
      }
      lds_barrier();   // every lane has read its inputs from hT
#pragma unroll
      for (int e = 0; e < 16; ++e) hT[n][(e & 3) + 8 * (e >> 2) + 4 * lk] = y[e];
      // (the barrier at the head of the next layer publishes them)
    }
  }
}

// backward of the activation-gradient chain, layers nl-1 .. 0 (weight gradients: grouped TN GEMMs over the stashed dth).
//   dT = g (H - x) T (1 - T);  dH = g T [H > 0];  g' = [dT | dH] . [Wt ; Wh]^T + g (1 - T)
// wT[l] is the (256, 128) transposed pair [Wt^T ; Wh^T] built by prepare_transposes.  grid = ceil(M / 32), block = 256,
// dynamic LDS (kHwBwdSmem bytes).
// AD (round 6, multi-speaker encoder): g' is the gradient of the layer's ADAPTER output x_l; it is stashed in dhx[l] (operand of the
// adapter's weight gradient and of the per-sequence bias sums) and taken through the adapter, g'' = g' . Wa_l[:128]^T (waT[l], one more
// 128 x 128 GEMM stage of two chunks), before it becomes the g of the layer below.
constexpr size_t kHwBwdSmem = sizeof(float) * ((size_t)2 * HC * HPAD + (size_t)HC * HPAD);
template <bool AD>
__global__ __launch_bounds__(256, 2) void highway_stack_bwd_kernel(HighwayStackBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float hw_smem[];
  float (*dT)[HPAD] = reinterpret_cast<float (*)[HPAD]>(hw_smem);                                  // [256][33] d[T|H], k-major
  float (*gT)[HPAD] = reinterpret_cast<float (*)[HPAD]>(hw_smem + 2 * HC * HPAD);                  // [128][33] dL/dy of the layer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * HB;
  const int n = 32 * wave + li;
  // element mapping of the pre-processing pass: thread -> (row pr + 8*i, 4 columns at pc4*4)
  const int pr = tid >> 5, pc4 = tid & 31;
  // incoming gradient tile -> gT
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = pr + 8 * i, m = m0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.M) v = *reinterpret_cast<const float4*>(a.g + (int64_t)m * HC + pc4 * 4);
    gT[pc4 * 4 + 0][r] = v.x;
    gT[pc4 * 4 + 1][r] = v.y;
    gT[pc4 * 4 + 2][r] = v.z;
    gT[pc4 * 4 + 3][r] = v.w;
  }
  // B fragments of a chunk: wT[l][k = 2 (32 c + j) + lk][n] (gate / candidate chunks cc < 4), waT[l][...] (adapter chunks cc = 4, 5)
  constexpr int CPL = AD ? 6 : 4;
  float w_r[2][BCK];
  auto load_chunk = [&](float (&w)[BCK], int l, int cc) {
    const float* src = (AD && cc >= 4) ? a.waT[l] + (2 * (cc - 4) * BCK + lk) * HC + n : a.wT[l] + (2 * cc * BCK + lk) * HC + n;
#pragma unroll
    for (int j = 0; j < BCK; ++j) w[j] = src[2 * j * HC];
  };
  const int nl = a.nl;
  load_chunk(w_r[0], nl - 1, 0);
  lds_barrier();

  f32x16 acc[2];
  // chunk i = (layer 3 - i / CPL, chunk i % CPL); layers above nl - 1 do not exist (CPL is even: a skipped layer does not change
  // the parity of the register sets, so the first executed chunk reads set 0, which the load above filled).
  static_assert(CPL % 2 == 0, "register-set parity relies on an even number of chunks per layer");
  hw_static_for<0, 4 * CPL>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int l = 3 - i / CPL, cc = i % CPL;
    if (l >= nl) return;
    constexpr int par = i & 1;
    constexpr bool hw = cc < 4;
    constexpr int c = hw ? cc : cc - 4;
    if (hw && c == 0) {
      // ---- pre-processing: d[T|H] of this layer from g, the stashed gates and the layer input ----
      // (all twelve loads first: issued between the stores of the previous rows they would have to queue behind them -- the
      //  compiler cannot prove that dth does not alias th / x, and vector-memory operations complete in order)
      float4 t4v[4], h4v[4], x4v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + pr + 8 * q;
        t4v[q] = h4v[q] = x4v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < a.M) {
          t4v[q] = *reinterpret_cast<const float4*>(a.th[l] + (int64_t)m * 2 * HC + pc4 * 4);
          h4v[q] = *reinterpret_cast<const float4*>(a.th[l] + (int64_t)m * 2 * HC + HC + pc4 * 4);
          x4v[q] = *reinterpret_cast<const float4*>(a.x[l] + (int64_t)m * HC + pc4 * 4);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = pr + 8 * q, m = m0 + r;
        const float4 t4 = t4v[q], h4 = h4v[q], x4 = x4v[q];
        const float tt[4] = {t4.x, t4.y, t4.z, t4.w}, hh[4] = {h4.x, h4.y, h4.z, h4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
        float dt[4], dh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = gT[pc4 * 4 + j][r];
          dt[j] = g * (hh[j] - xx[j]) * tt[j] * (1.f - tt[j]);
          dh[j] = hh[j] > 0.f ? g * tt[j] : 0.f;
          dT[pc4 * 4 + j][r] = dt[j];
          dT[HC + pc4 * 4 + j][r] = dh[j];
        }
        if (m < a.M) {
          *reinterpret_cast<float4*>(a.dth[l] + (int64_t)m * 2 * HC + pc4 * 4) = make_float4(dt[0], dt[1], dt[2], dt[3]);
          *reinterpret_cast<float4*>(a.dth[l] + (int64_t)m * 2 * HC + HC + pc4 * 4) = make_float4(dh[0], dh[1], dh[2], dh[3]);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
      lds_barrier();
    }
    if (!hw && c == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;   // (gT holds g' since the barrier behind the gate stage's epilogue)
    }
    // the next chunk's weights fly under this chunk's MFMAs (the last chunk of a layer fetches the first of the layer below)
    if constexpr (i + 1 < 4 * CPL) load_chunk(w_r[par ^ 1], 3 - (i + 1) / CPL, (i + 1) % CPL);
    __builtin_amdgcn_sched_barrier(0);   // the prefetch stays HERE (see the forward kernel)
    // ---- gate stage: g' = d[T|H] . wT (K = 256); adapter stage: g'' = g' . waT (K = 128) ----
#pragma unroll
    for (int hhalf = 0; hhalf < 2; ++hhalf) {
      float av[BCK / 2];
#pragma unroll
      for (int j = 0; j < BCK / 2; ++j) {
        const int k = 2 * (c * BCK + hhalf * (BCK / 2) + j) + lk;
        av[j] = hw ? dT[k][li] : gT[k][li];
      }
#pragma unroll
      for (int j = 0; j < BCK / 2; ++j)
        acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], w_r[par][hhalf * (BCK / 2) + j], acc[j & 1], 0, 0, 0);
    }
    if (hw && c == 3) {
      // ---- epilogue: + g (1 - T); the result is the gradient of the layer's input (AD: of its adapter's output) ----
      float gn[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
        const int m = m0 + r;
        const float t = m < a.M ? a.th[l][(int64_t)m * 2 * HC + n] : 0.f;
        gn[e] = acc[0][e] + acc[1][e] + gT[n][r] * (1.f - t);
      }
      lds_barrier();   // every lane has read the old g (and every wave is done with dT)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
        gT[n][r] = gn[e];
        if (AD) {
          if (m0 + r < a.M) a.dhx[l][(int64_t)(m0 + r) * HC + n] = gn[e];
        } else if (l == 0 && m0 + r < a.M) {
          a.gout[(int64_t)(m0 + r) * HC + n] = gn[e];
        }
      }
      lds_barrier();
    }
    if (!hw && c == 1) {
      // ---- adapter epilogue: g'' is the g of the layer below (layer 0: the CBHG's residual gradient) ----
      lds_barrier();   // every wave has read g' from gT
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
        const float v = acc[0][e] + acc[1][e];
        gT[n][r] = v;
        if (l == 0 && m0 + r < a.M) a.gout[(int64_t)(m0 + r) * HC + n] = v;
      }
      lds_barrier();
    }
  });
}

}  // namespace

int launch_highway_stack_bwd(const HighwayStackBwdArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.nl >= 1 && a.nl <= 4 && a.g && a.gout, "highway_stack_bwd: bad arguments");
  const bool ad = a.waT[0] != nullptr;
  TACO_REQUIRE(!ad || (a.nl == 4 && a.waT[1] && a.waT[2] && a.waT[3] && a.dhx[0] && a.dhx[1] && a.dhx[2] && a.dhx[3]),
               "highway_stack_bwd: the adapter form needs waT / dhx of all four layers");
  static DynSmemOnce once, once_ad;
  TACO_REQUIRE(ensure_dyn_smem(ad ? once_ad : once, ad ? reinterpret_cast<const void*>(highway_stack_bwd_kernel<true>)
                                                       : reinterpret_cast<const void*>(highway_stack_bwd_kernel<false>), kHwBwdSmem),
               "highway_stack_bwd: cannot reserve %zu bytes of LDS", kHwBwdSmem);
  const int pslot = taco_prof_begin(2, s);
  taco_prof_label(2, pslot, "highway-bwd M=%d nl=%d%s", a.M, a.nl, ad ? " +adapters" : "");
  if (ad) TACO_KLAUNCH(highway_stack_bwd_kernel<true>, dim3(cdiv(a.M, HB)), dim3(256), kHwBwdSmem, s, a);
  else TACO_KLAUNCH(highway_stack_bwd_kernel<false>, dim3(cdiv(a.M, HB)), dim3(256), kHwBwdSmem, s, a);
  taco_prof_end(2, pslot, s, 2.0 * a.M * HC * (ad ? 3 : 2) * HC * a.nl);
  TACO_LAUNCH_CHECK("highway_stack_bwd");
  return TACO_OK;
}

int launch_highway_stack_fwd(const HighwayStackArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.nl >= 1 && a.nl <= 4 && a.x, "highway_stack_fwd: bad arguments");
  const bool ad = a.wa[0] != nullptr;
  TACO_REQUIRE(!ad || (a.nl == 4 && a.T > 0 && a.wa[1] && a.wa[2] && a.wa[3] && a.rowb[0] && a.rowb[1] && a.rowb[2] && a.rowb[3] &&
                       a.hx[0] && a.hx[1] && a.hx[2] && a.hx[3]),
               "highway_stack_fwd: the adapter form needs wa / rowb / hx of all four layers and the sequence length");
  const int pslot = taco_prof_begin(2, s);
  taco_prof_label(2, pslot, "highway-fwd M=%d nl=%d%s", a.M, a.nl, ad ? " +adapters" : "");
  if (ad) TACO_KLAUNCH(highway_stack_fwd_kernel<true>, dim3(cdiv(a.M, HB)), dim3(256), 0, s, a);
  else TACO_KLAUNCH(highway_stack_fwd_kernel<false>, dim3(cdiv(a.M, HB)), dim3(256), 0, s, a);
  taco_prof_end(2, pslot, s, 2.0 * a.M * HC * (ad ? 3 : 2) * HC * a.nl);
  TACO_LAUNCH_CHECK("highway_stack_fwd");
  return TACO_OK;
}
