// highway.hip -- the CBHG's four highway layers (ops.py:27-46, 97-107) as ONE kernel per direction of the pass.
//
// The layers are row-wise: y = relu(x Wh + bh) * sigmoid(x Wt + bt) + x * (1 - sigmoid(x Wt + bt)), y feeds the next layer.
// Run as separate GEMMs each layer is a 0.4-0.75 GFLOP launch with K = 128 that cannot fill the chip (16-26 TF measured)
// plus an elementwise blend; here a workgroup keeps a 32-row tile resident in LDS and walks all layers, streaming the
// (128 x 256) [Wt | Wh] of each layer through LDS in k-tiles.  MFMA tiling: 4 waves, wave w owns the 32 gate columns
// [32w, 32w+32) AND the same 32 candidate columns (two 32x32 accumulators), so T and H of an element meet in one lane and
// the blend is the epilogue.  The backward kernel does the same for the activation-gradient chain; the weight gradients
// stay grouped TN GEMMs over the stashed d[T|H] (model.hip).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int HB = 32;         // rows per workgroup
constexpr int HC = 128;        // highway width
constexpr int HK = 16;         // k-tile
constexpr int HPAD = HB + 1;   // hT row pitch (conflict-free column reads)
constexpr int WPITCH = 2 * HC + 8;

// forward: grid = ceil(M / 32), block = 256
__global__ __launch_bounds__(256) void highway_stack_fwd_kernel(HighwayStackArgs a) {
  __shared__ __attribute__((aligned(16))) float hT[HC][HPAD];          // current layer input, feature-major
  __shared__ __attribute__((aligned(16))) float Ws[2][HK][WPITCH];     // k-tile of [Wt | Wh]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * HB;

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
  // W loader: 16 rows x 64 float4 per k-tile = 4 float4 per thread
  const int w_c4 = tid & 63, w_r = tid >> 6;   // rows w_r + 4*i
  // two register sets: a k-tile is fetched TWO iterations before it is written to LDS (one iteration of 16 MFMAs is shorter
  // than an L2 round trip)
  float4 rw[2][4];
  auto load_w = [&](float4 (&r)[4], const float* wt, const float* wh, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + w_r + 4 * i;
      const float* src = w_c4 < 32 ? wt + (int64_t)k * HC + w_c4 * 4 : wh + (int64_t)k * HC + (w_c4 - 32) * 4;
      r[i] = *reinterpret_cast<const float4*>(src);
    }
  };
  auto store_w = [&](const float4 (&r)[4], int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Ws[buf][w_r + 4 * i][w_c4 * 4]) = r[i];
  };

#pragma unroll
  for (int l = 0; l < 4; ++l) {   // (unrolled: the per-layer pointer arrays of the argument struct stay in registers)
    if (l >= a.nl) break;
    const float* wt = a.wt[l];
    const float* wh = a.wh[l];
    f32x16 accT, accH;
#pragma unroll
    for (int e = 0; e < 16; ++e) accT[e] = accH[e] = 0.f;
    constexpr int NKT = HC / HK;
    load_w(rw[0], wt, wh, 0);
    store_w(rw[0], 0);
    load_w(rw[1], wt, wh, HK);        // tile 1 -> set 1
    load_w(rw[0], wt, wh, 2 * HK);    // tile 2 -> set 0
    __syncthreads();   // also publishes hT (first layer: the input tile; later layers: the previous layer's output)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int buf = kt & 1;
#pragma unroll
      for (int kk = 0; kk < HK; kk += 2) {
        const float av = hT[kt * HK + kk + lk][li];
        const float bt = Ws[buf][kk + lk][32 * wave + li];
        const float bh = Ws[buf][kk + lk][HC + 32 * wave + li];
        accT = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bt, accT, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bh, accH, 0, 0, 0);
      }
      if (kt + 1 < NKT) store_w(rw[(kt + 1) & 1], buf ^ 1);                       // tile kt+1 (fetched two iterations ago)
      if (kt + 3 < NKT) load_w(rw[(kt + 1) & 1], wt, wh, (kt + 3) * HK);          // tile kt+3 into the freed set
      __syncthreads();
    }
    // epilogue: C layout col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int n = 32 * wave + li;
    const float bT = a.bt[l][n], bH = a.bh[l][n];
    float y[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
      const int m = m0 + r;
      const float T = sigmoid_f(accT[e] + bT);
      const float H = fmaxf(accH[e] + bH, 0.f);
      const float h = hT[n][r];
      y[e] = H * T + h * (1.f - T);
      if (m < a.M) {
        a.th[l][(int64_t)m * 2 * HC + n] = T;
        a.th[l][(int64_t)m * 2 * HC + HC + n] = H;
        a.y[l][(int64_t)m * HC + n] = y[e];
      }
    }
    __syncthreads();   // every lane has read its inputs from hT
#pragma unroll
    for (int e = 0; e < 16; ++e) hT[n][(e & 3) + 8 * (e >> 2) + 4 * lk] = y[e];
    // (the barrier at the top of the next layer's k-loop publishes them)
  }
}

// backward of the activation-gradient chain, layers nl-1 .. 0 (weight gradients: grouped TN GEMMs over the stashed dth).
//   dT = g (H - x) T (1 - T);  dH = g T [H > 0];  g' = [dT | dH] . [Wt ; Wh]^T + g (1 - T)
// wT[l] is the (256, 128) transposed pair [Wt^T ; Wh^T] built by prepare_transposes.  grid = ceil(M / 32), block = 256,
// dynamic LDS (kHwBwdSmem bytes).
constexpr int WBP = HC + 4;   // pitch of the (16 x 128) k-tile of wT
constexpr size_t kHwBwdSmem = sizeof(float) * ((size_t)2 * HC * HPAD + (size_t)HC * HPAD + (size_t)2 * HK * WBP);
__global__ __launch_bounds__(256) void highway_stack_bwd_kernel(HighwayStackBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float hw_smem[];
  float (*dT)[HPAD] = reinterpret_cast<float (*)[HPAD]>(hw_smem);                                  // [256][33] d[T|H], k-major
  float (*gT)[HPAD] = reinterpret_cast<float (*)[HPAD]>(hw_smem + 2 * HC * HPAD);                  // [128][33] dL/dy of the layer
  float (*Ws)[HK][WBP] = reinterpret_cast<float (*)[HK][WBP]>(hw_smem + 3 * HC * HPAD);            // [2][16][132]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * HB;
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
  const int w_c4 = tid & 31, w_r = tid >> 5;   // k-tile loader: 16 rows x 32 float4 = 2 float4 per thread
  float4 rw[2][2];   // two register sets, tiles fetched two iterations ahead (see the forward kernel)
  auto load_w = [&](float4 (&r)[2], const float* w, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) r[i] = *reinterpret_cast<const float4*>(w + (int64_t)(k0 + w_r + 8 * i) * HC + w_c4 * 4);
  };
  auto store_w = [&](const float4 (&r)[2], int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&Ws[buf][w_r + 8 * i][w_c4 * 4]) = r[i];
  };
  __syncthreads();

#pragma unroll
  for (int l = 3; l >= 0; --l) {
    if (l >= a.nl) continue;
    const float* w = a.wT[l];
    load_w(rw[0], w, 0);
    // ---- pre-processing: d[T|H] of this layer from g, the stashed gates and the layer input ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = pr + 8 * i, m = m0 + r;
      float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f), h4 = t4, x4 = t4;
      if (m < a.M) {
        t4 = *reinterpret_cast<const float4*>(a.th[l] + (int64_t)m * 2 * HC + pc4 * 4);
        h4 = *reinterpret_cast<const float4*>(a.th[l] + (int64_t)m * 2 * HC + HC + pc4 * 4);
        x4 = *reinterpret_cast<const float4*>(a.x[l] + (int64_t)m * HC + pc4 * 4);
      }
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
    store_w(rw[0], 0);
    load_w(rw[1], w, HK);
    load_w(rw[0], w, 2 * HK);
    __syncthreads();
    // ---- g' = d[T|H] . wT  (K = 256) ----
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int NKT = 2 * HC / HK;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int buf = kt & 1;
#pragma unroll
      for (int kk = 0; kk < HK; kk += 2) {
        const float av = dT[kt * HK + kk + lk][li];
        const float bv = Ws[buf][kk + lk][32 * wave + li];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      if (kt + 1 < NKT) store_w(rw[(kt + 1) & 1], buf ^ 1);
      if (kt + 3 < NKT) load_w(rw[(kt + 1) & 1], w, (kt + 3) * HK);
      __syncthreads();
    }
    // ---- epilogue: + g (1 - T); the result is the next (lower) layer's g ----
    const int n = 32 * wave + li;
    float gn[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
      const int m = m0 + r;
      const float t = m < a.M ? a.th[l][(int64_t)m * 2 * HC + n] : 0.f;
      gn[e] = acc[e] + gT[n][r] * (1.f - t);
    }
    __syncthreads();   // every lane has read the old g
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
      gT[n][r] = gn[e];
      if (l == 0 && m0 + r < a.M) a.gout[(int64_t)(m0 + r) * HC + n] = gn[e];
    }
    __syncthreads();
  }
}

}  // namespace

int launch_highway_stack_bwd(const HighwayStackBwdArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.nl >= 1 && a.nl <= 4 && a.g && a.gout, "highway_stack_bwd: bad arguments");
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(highway_stack_bwd_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHwBwdSmem) == hipSuccess;
  TACO_REQUIRE(ok, "highway_stack_bwd: cannot reserve %zu bytes of LDS", kHwBwdSmem);
  const int pslot = taco_prof_begin(2, s);
  hipLaunchKernelGGL(highway_stack_bwd_kernel, dim3(cdiv(a.M, HB)), dim3(256), kHwBwdSmem, s, a);
  taco_prof_end(2, pslot, s, 2.0 * a.M * HC * 2 * HC * a.nl);
  TACO_LAUNCH_CHECK("highway_stack_bwd");
  return TACO_OK;
}

int launch_highway_stack_fwd(const HighwayStackArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.nl >= 1 && a.nl <= 4 && a.x, "highway_stack_fwd: bad arguments");
  const int pslot = taco_prof_begin(2, s);
  hipLaunchKernelGGL(highway_stack_fwd_kernel, dim3(cdiv(a.M, HB)), dim3(256), 0, s, a);
  taco_prof_end(2, pslot, s, 2.0 * a.M * HC * 2 * HC * a.nl);
  TACO_LAUNCH_CHECK("highway_stack_fwd");
  return TACO_OK;
}
