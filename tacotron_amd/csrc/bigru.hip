// bigru.hip -- persistent recurrent kernels for the CBHG bidirectional GRU(128) (ops.py:117-128;
// tf.nn.bidirectional_dynamic_rnn(GRUCell(128), GRUCell(128), h) with no sequence_length).
//
// The x-side projections of all T steps are hoisted into one MFMA GEMM (gemm.hip); only the h-side recurrence
// runs here.  One workgroup (256 threads = 4 waves, one per SIMD) owns ONE (direction, batch row) for the whole
// sequence: the h-side weights Wg[128:,:] (128x256) and Wc[128:,:] (128x128) -- 48K floats -- live in that
// workgroup's VGPRs (192 per lane) for all T steps, the 128-float hidden state lives in LDS, and nothing but the
// per-step x-projection (1.5 KB) and the outputs touch HBM inside the loop.  No inter-workgroup communication.
//
// GRUCell r1.2: [r,u] = sigmoid([x,h].Wg + bg); c = tanh([x, r*h].Wc + bc); h' = u*h + (1-u)*c.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int H = kCb;  // 128

// grid = (B, 2 directions); block = 256
__global__ __launch_bounds__(256, 1) void bigru_fwd_kernel(const float* __restrict__ xg, BiGruWeights w,
                                                           const float* __restrict__ h0, float* __restrict__ out,
                                                           float* __restrict__ ruc, int B, int T) {
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  const int col = j & (H - 1), half = j >> 7;
  __shared__ __attribute__((aligned(16))) float hs[H];
  __shared__ __attribute__((aligned(16))) float rhs[H];
  __shared__ float us[H];
  __shared__ float cpart[H];

  // h-side weights -> registers
  float wgh[H];
  float wch[H / 2];
  {
    const float* wg = w.wg[d] + (int64_t)H * (2 * H) + j;  // rows 128.., column j
#pragma unroll
    for (int k = 0; k < H; ++k) wgh[k] = wg[(int64_t)k * (2 * H)];
    const float* wc = w.wc[d] + (int64_t)(H + half * (H / 2)) * H + col;
#pragma unroll
    for (int k = 0; k < H / 2; ++k) wch[k] = wc[(int64_t)k * H];
  }
  if (j < H) hs[j] = h0 ? h0[(int64_t)b * H + j] : 0.f;   // initial_state_fw = initial_state_bw = s (ops.py:123-124)
  lds_barrier();

  const int64_t row0 = (int64_t)b * T;
  const int tstart = d == 0 ? 0 : T - 1, tstep = d == 0 ? 1 : -1;
  float xg_g = xg[(row0 + tstart) * (6 * H) + d * 3 * H + j];
  float xg_c = half == 0 ? xg[(row0 + tstart) * (6 * H) + d * 3 * H + 2 * H + col] : 0.f;

  for (int s = 0, t = tstart; s < T; ++s, t += tstep) {
    // prefetch next step's x-projection (independent of the recurrence)
    float nxg_g = 0.f, nxg_c = 0.f;
    if (s + 1 < T) {
      const int64_t nr = (row0 + t + tstep) * (6 * H) + d * 3 * H;
      nxg_g = xg[nr + j];
      if (half == 0) nxg_c = xg[nr + 2 * H + col];
    }
    // ---- gates: thread j owns gate column j ----
    float acc0 = xg_g, acc1 = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < H / 4; ++k4) {
      const float4 hv = reinterpret_cast<const float4*>(hs)[k4];
      acc0 = fmaf(hv.x, wgh[4 * k4 + 0], acc0);
      acc1 = fmaf(hv.y, wgh[4 * k4 + 1], acc1);
      acc0 = fmaf(hv.z, wgh[4 * k4 + 2], acc0);
      acc1 = fmaf(hv.w, wgh[4 * k4 + 3], acc1);
    }
    const float g = sigmoid_f(acc0 + acc1);
    const float hprev = hs[col];
    if (half == 0) rhs[col] = g * hprev;   // r * h
    else us[col] = g;                      // u
    lds_barrier();
    // ---- candidate: thread (col, half) reduces its half of r*h ----
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < H / 8; ++k4) {
      const float4 rv = reinterpret_cast<const float4*>(rhs)[half * (H / 8) + k4];
      p0 = fmaf(rv.x, wch[4 * k4 + 0], p0);
      p1 = fmaf(rv.y, wch[4 * k4 + 1], p1);
      p0 = fmaf(rv.z, wch[4 * k4 + 2], p0);
      p1 = fmaf(rv.w, wch[4 * k4 + 3], p1);
    }
    if (half == 1) cpart[col] = p0 + p1;
    lds_barrier();
    if (half == 0) {
      const float c = tanh_f(xg_c + p0 + p1 + cpart[col]);
      const float u = us[col];
      const float hn = u * hprev + (1.f - u) * c;
      hs[col] = hn;
      out[(row0 + t) * (2 * H) + d * H + col] = hn;
      if (ruc) {
        float* rp = ruc + (row0 + t) * (6 * H) + d * 3 * H;
        rp[col] = g;          // r (this thread's gate column is r_col)
        rp[H + col] = u;
        rp[2 * H + col] = c;
      }
    }
    xg_g = nxg_g;
    xg_c = nxg_c;
    lds_barrier();
  }
}

// Backward recurrence.  grid = (B, 2); block = 256.
// Per step (reverse of the forward order), with h_prev = previous forward state, dh = carried gradient:
//   dht = dh + dout; du = dht*(h_prev - c); dc = dht*(1-u); dcp = dc*(1-c^2); d(rh) = dcp . Wc_h^T
//   dr = d(rh)*h_prev; dgp = [dr*r(1-r), du*u(1-u)]; dh = dht*u + d(rh)*r + dgp . Wg_h^T
__global__ __launch_bounds__(256, 1) void bigru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                           const float* __restrict__ ruc, BiGruBwdWeights w,
                                                           const float* __restrict__ h0, float* __restrict__ dxg,
                                                           float* __restrict__ rh_out, float* __restrict__ dh0, int B,
                                                           int T) {
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  const int col = j & (H - 1), half = j >> 7;
  __shared__ __attribute__((aligned(16))) float dcp_s[2][H];
  __shared__ __attribute__((aligned(16))) float dgp_s[2][2 * H];
  __shared__ float part_s[2][H];

  // transposed h-side weights -> registers.  wchT (128 [col], 128 [k]); wghT (256 [j], 128 [k]).
  float wc_r[H / 2];   // d(rh)[k=col] partial over cols in [half*64, half*64+64)
  float wg_r[H];       // dh[k=col] partial over gate columns in [half*128, half*128+128)
  {
    const float* p = w.wchT[d] + (int64_t)(half * (H / 2)) * H + col;
#pragma unroll
    for (int i = 0; i < H / 2; ++i) wc_r[i] = p[(int64_t)i * H];
    const float* q = w.wghT[d] + (int64_t)(half * H) * H + col;
#pragma unroll
    for (int i = 0; i < H; ++i) wg_r[i] = q[(int64_t)i * H];
  }

  const int64_t row0 = (int64_t)b * T;
  // forward order for d=0 is t = 0..T-1, so backward visits T-1..0; for d=1 forward is T-1..0, backward 0..T-1.
  const int tstart = d == 0 ? T - 1 : 0, tstep = d == 0 ? -1 : 1;
  float dh = 0.f;  // carried gradient for k = col (held by half 0)

  for (int s = 0, t = tstart; s < T; ++s, t += tstep) {
    const int buf = s & 1;
    const int tp = t + tstep;  // time index of the previous forward step (h_prev lives there)
    const bool has_prev = (s + 1 < T);
    float r = 0.f, u = 0.f, c = 0.f, hp = 0.f, dht = 0.f;
    if (half == 0) {
      const float* rp = ruc + (row0 + t) * (6 * H) + d * 3 * H;
      r = rp[col];
      u = rp[H + col];
      c = rp[2 * H + col];
      hp = has_prev ? out[(row0 + tp) * (2 * H) + d * H + col] : (h0 ? h0[(int64_t)b * H + col] : 0.f);
      dht = dh + dout[(row0 + t) * (2 * H) + d * H + col];
      const float du = dht * (hp - c);
      const float dc = dht * (1.f - u);
      const float dcp = dc * (1.f - c * c);
      const float dup = du * u * (1.f - u);
      dcp_s[buf][col] = dcp;
      dgp_s[buf][H + col] = dup;
      float* xo = dxg + (row0 + t) * (6 * H) + d * 3 * H;
      xo[2 * H + col] = dcp;
      xo[H + col] = dup;
      rh_out[(row0 + t) * (2 * H) + d * H + col] = r * hp;
    }
    lds_barrier();
    // d(rh)[col] partial
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int i4 = 0; i4 < H / 8; ++i4) {
      const float4 v = reinterpret_cast<const float4*>(dcp_s[buf])[half * (H / 8) + i4];
      p0 = fmaf(v.x, wc_r[4 * i4 + 0], p0);
      p1 = fmaf(v.y, wc_r[4 * i4 + 1], p1);
      p0 = fmaf(v.z, wc_r[4 * i4 + 2], p0);
      p1 = fmaf(v.w, wc_r[4 * i4 + 3], p1);
    }
    if (half == 1) part_s[0][col] = p0 + p1;
    lds_barrier();
    float dh_acc = 0.f;
    if (half == 0) {
      const float drh = p0 + p1 + part_s[0][col];
      const float dr = drh * hp;
      const float drp = dr * r * (1.f - r);
      dgp_s[buf][col] = drp;
      dxg[(row0 + t) * (6 * H) + d * 3 * H + col] = drp;
      dh_acc = dht * u + drh * r;
    }
    lds_barrier();
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i4 = 0; i4 < H / 4; ++i4) {
      const float4 v = reinterpret_cast<const float4*>(dgp_s[buf])[half * (H / 4) + i4];
      q0 = fmaf(v.x, wg_r[4 * i4 + 0], q0);
      q1 = fmaf(v.y, wg_r[4 * i4 + 1], q1);
      q0 = fmaf(v.z, wg_r[4 * i4 + 2], q0);
      q1 = fmaf(v.w, wg_r[4 * i4 + 3], q1);
    }
    if (half == 1) part_s[1][col] = q0 + q1;
    lds_barrier();
    if (half == 0) dh = dh_acc + q0 + q1 + part_s[1][col];
  }
  if (dh0 && half == 0) dh0[((int64_t)d * B + b) * H + col] = dh;   // gradient w.r.t. the initial state
}

}  // namespace

int launch_bigru_fwd(const float* xg, const BiGruWeights& w, const float* h0, float* out, float* ruc, int B, int T,
                     hipStream_t s) {
  TACO_REQUIRE(B > 0 && T > 0, "bigru_fwd: bad dims");
  hipLaunchKernelGGL(bigru_fwd_kernel, dim3(B, 2), dim3(256), 0, s, xg, w, h0, out, ruc, B, T);
  TACO_LAUNCH_CHECK("bigru_fwd");
  return TACO_OK;
}

int launch_bigru_bwd(const float* dout, const float* out, const float* ruc, const BiGruBwdWeights& w, const float* h0,
                     float* dxg, float* rh, float* dh0, int B, int T, hipStream_t s) {
  TACO_REQUIRE(B > 0 && T > 0, "bigru_bwd: bad dims");
  hipLaunchKernelGGL(bigru_bwd_kernel, dim3(B, 2), dim3(256), 0, s, dout, out, ruc, w, h0, dxg, rh, dh0, B, T);
  TACO_LAUNCH_CHECK("bigru_bwd");
  return TACO_OK;
}
