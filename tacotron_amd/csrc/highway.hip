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
  float4 rw[4];
  auto load_w = [&](const float* wt, const float* wh, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + w_r + 4 * i;
      const float* src = w_c4 < 32 ? wt + (int64_t)k * HC + w_c4 * 4 : wh + (int64_t)k * HC + (w_c4 - 32) * 4;
      rw[i] = *reinterpret_cast<const float4*>(src);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Ws[buf][w_r + 4 * i][w_c4 * 4]) = rw[i];
  };

  for (int l = 0; l < a.nl; ++l) {
    const float* wt = a.wt[l];
    const float* wh = a.wh[l];
    f32x16 accT, accH;
#pragma unroll
    for (int e = 0; e < 16; ++e) accT[e] = accH[e] = 0.f;
    load_w(wt, wh, 0);
    store_w(0);
    __syncthreads();   // also publishes hT (first layer: the input tile; later layers: the previous layer's output)
    constexpr int NKT = HC / HK;
    for (int kt = 0; kt < NKT; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < NKT) load_w(wt, wh, (kt + 1) * HK);
#pragma unroll
      for (int kk = 0; kk < HK; kk += 2) {
        const float av = hT[kt * HK + kk + lk][li];
        const float bt = Ws[buf][kk + lk][32 * wave + li];
        const float bh = Ws[buf][kk + lk][HC + 32 * wave + li];
        accT = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bt, accT, 0, 0, 0);
        accH = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bh, accH, 0, 0, 0);
      }
      if (kt + 1 < NKT) store_w(buf ^ 1);
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

}  // namespace

int launch_highway_stack_fwd(const HighwayStackArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.nl >= 1 && a.nl <= 4 && a.x, "highway_stack_fwd: bad arguments");
  hipLaunchKernelGGL(highway_stack_fwd_kernel, dim3(cdiv(a.M, HB)), dim3(256), 0, s, a);
  TACO_LAUNCH_CHECK("highway_stack_fwd");
  return TACO_OK;
}
