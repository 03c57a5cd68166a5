// bigru.hip -- persistent recurrent kernels for the CBHG bidirectional GRU(128) (ops.py:117-128;
// tf.nn.bidirectional_dynamic_rnn(GRUCell(128), GRUCell(128), h) with no sequence_length).
//
// The x-side projections of all T steps are hoisted into one MFMA GEMM (gemm.hip); only the h-side recurrence
// runs here.  One workgroup (512 threads = 8 waves, two per SIMD) owns ONE (direction, batch row) for the whole
// sequence: the h-side weights Wg[128:,:] (128x256) and Wc[128:,:] (128x128) -- 48K floats -- live in that
// workgroup's VGPRs (96 per lane) for all T steps, the 128-float hidden state lives in LDS, the per-step inputs arrive
// by direct global->LDS DMA in 8-step chunks, and only the outputs leave the CU inside the loop.  Every reduction is
// split over the four lanes of a quad and combined with DPP quad_perm swaps (no LDS partials, 2 barriers per step).
// No inter-workgroup communication.
//
// GRUCell r1.2: [r,u] = sigmoid([x,h].Wg + bg); c = tanh([x, r*h].Wc + bc); h' = u*h + (1-u)*c.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int H = kCb;  // 128

// Direct global->LDS DMA of 64 consecutive floats (one dword per lane): LDS destination = wave-uniform base + lane*4.
// No VGPR destination, so nothing for the compiler to wait on; the issuing wave waits with an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void dma64(const float* gsrc_lane, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// sum over the 4 lanes of a quad (every lane gets the total): two DPP quad_perm swaps, no LDS
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_move<0xb1>(0.f, v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4e>(0.f, v);   // quad_perm [2,3,0,1]
  return v;
}

constexpr int NTG = 512;
// The four k-quarters of a quad read their ds_read_b128 slices of an LDS vector at the same time; unpadded, the quarter
// bases are 128 B (32-float vectors) or 256 B (64-float) apart and fall on the same banks (PMC: LDS_BANK_CONFLICT on 12 % /
// 25 % of the forward / backward wave cycles).  Each quarter therefore gets 4 floats of padding: base = q * (len + 4).
__device__ __forceinline__ int pad32(int c) { return c + 4 * (c >> 5); }   // 32-float quarters (128-wide vectors)
__device__ __forceinline__ int pad64(int c) { return c + 4 * (c >> 6); }   // 64-float quarters (256-wide vectors)
constexpr int HP = H + 16;         // padded 128-vector
constexpr int H2P = 2 * H + 16;    // padded 256-vector
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }      // 8 waves = 2 per SIMD: a lone wave per SIMD only issues ~1 instruction per 4 cycles
constexpr int CH = 8;         // time steps per DMA chunk

// Forward.  grid = (B, 2 directions); block = 512.
//  gates    : thread (cp = t>>2, kq = t&3) owns gate columns {cp, cp+128} (= r_cp and u_cp) over k in [32kq, 32kq+32):
//             64 weights in VGPRs, 8 broadcast ds_read_b128 of h, quad_sum combines the four k-quarters in-register.
//  candidate: thread (cc = t>>2, kq) owns candidate column cc over the same k-quarter (32 weights); lane kq==0 of each
//             quad finishes c, h' and issues the stores.  Two workgroup barriers per step.
//  inputs   : the hoisted x-projections arrive by LDS-DMA in chunks of CH steps, issued by wave 7 one chunk ahead.
__global__ __launch_bounds__(NTG, 2) void bigru_fwd_kernel(const float* __restrict__ xg, BiGruWeights w,
                                                           const float* __restrict__ h0, float* __restrict__ out,
                                                           float* __restrict__ ruc, int B, int T, long long* trace) {
  (void)trace;
  const int b = blockIdx.x, d = blockIdx.y, t_ = threadIdx.x;
  const int cp = t_ >> 2, kq = t_ & 3, lane = t_ & 63, wv = t_ >> 6;
  __shared__ __attribute__((aligned(16))) float hs[HP];
  __shared__ __attribute__((aligned(16))) float rhs[HP];
  __shared__ __attribute__((aligned(16))) float xgs[2][CH][3 * H];

  // Wg_h[32kq + k][cp], Wg_h[..][128 + cp], Wc_h[32kq + k][cp], held as k-pairs so the dot products run on v_pk_fma_f32
  f2 wr[H / 8], wu[H / 8], wcand[H / 8];
  {
    const float* wg = w.wg[d] + (int64_t)(H + kq * (H / 4)) * (2 * H) + cp;
    const float* wc = w.wc[d] + (int64_t)(H + kq * (H / 4)) * H + cp;
#pragma unroll
    for (int k = 0; k < H / 8; ++k) {
      wr[k] = f2{wg[(int64_t)(2 * k) * (2 * H)], wg[(int64_t)(2 * k + 1) * (2 * H)]};
      wu[k] = f2{wg[(int64_t)(2 * k) * (2 * H) + H], wg[(int64_t)(2 * k + 1) * (2 * H) + H]};
      wcand[k] = f2{wc[(int64_t)(2 * k) * H], wc[(int64_t)(2 * k + 1) * H]};
    }
  }
  if (t_ < H) hs[pad32(t_)] = h0 ? h0[(int64_t)b * H + t_] : 0.f;   // initial_state_fw = initial_state_bw = s (ops.py:123-124)

  const int64_t row0 = (int64_t)b * T;
  const int tstart = d == 0 ? 0 : T - 1, tstep = d == 0 ? 1 : -1;
  // wave i DMAs step i of chunk c (recurrence steps [c*CH, c*CH+CH)) into xgs[c & 1][i]   (CH == number of waves)
  auto dma_chunk = [&](int c) {
    const int s = c * CH + wv;
    if (s < T) {
      const float* src = xg + (row0 + tstart + s * tstep) * (6 * H) + d * 3 * H + lane;
      float* dst = &xgs[c & 1][wv][0];
#pragma unroll
      for (int q = 0; q < 6; ++q) dma64(src + 64 * q, dst + 64 * q);
    }
  };
  dma_chunk(0);
  wait_vm0();
  lds_barrier();

  for (int s = 0, t = tstart; s < T; ++s, t += tstep) {
    const int c = s / CH, i = s - c * CH;
    if (i == 0 && (c + 1) * CH < T) dma_chunk(c + 1);
    const float* xrow = &xgs[c & 1][i][0];
    // ---- gates ----
    f2 a0 = {0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) {
      const float4 hv = reinterpret_cast<const float4*>(hs + kq * (H / 4 + 4))[k4];
      const f2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
      a0 = pk_fma(h01, wr[2 * k4], a0); b0 = pk_fma(h01, wu[2 * k4], b0);
      a1 = pk_fma(h23, wr[2 * k4 + 1], a1); b1 = pk_fma(h23, wu[2 * k4 + 1], b1);
    }
    const float rg = sigmoid_fast(quad_sum(hsum(a0 + a1)) + xrow[cp]);
    const float ug = sigmoid_fast(quad_sum(hsum(b0 + b1)) + xrow[H + cp]);
    const float hprev = hs[pad32(cp)];
    if (kq == 0) {
      rhs[pad32(cp)] = rg * hprev;
    }
    lds_barrier();
    // ---- candidate ----
    f2 p0 = {0.f, 0.f}, p1 = p0;
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) {
      const float4 rv = reinterpret_cast<const float4*>(rhs + kq * (H / 4 + 4))[k4];
      p0 = pk_fma(f2{rv.x, rv.y}, wcand[2 * k4], p0);
      p1 = pk_fma(f2{rv.z, rv.w}, wcand[2 * k4 + 1], p1);
    }
    const float cpre = quad_sum(hsum(p0 + p1)) + xrow[2 * H + cp];
    // every wave's share of the next chunk has had CH-1 steps to land; wait before this step's (younger) stores are issued
    if (i == CH - 1) wait_vm0();
    if (kq == 0) {
      const float cc = tanh_fast(cpre);
      const float hn = ug * hprev + (1.f - ug) * cc;
      hs[pad32(cp)] = hn;
      out[(row0 + t) * (2 * H) + d * H + cp] = hn;
      if (ruc) {
        float* rp = ruc + (row0 + t) * (6 * H) + d * 3 * H;
        rp[cp] = rg;
        rp[H + cp] = ug;
        rp[2 * H + cp] = cc;
      }
    }
    lds_barrier();
  }
}

// Backward recurrence.  grid = (B, 2); block = 512.
// Per step (reverse of the forward order), with h_prev = previous forward state, dh = carried gradient:
//   dht = dh + dout; du = dht*(h_prev - c); dc = dht*(1-u); dcp = dc*(1-c^2); d(rh) = dcp . Wc_h^T
//   dr = d(rh)*h_prev; dgp = [dr*r(1-r), du*u(1-u)]; dh = dht*u + d(rh)*r + dgp . Wg_h^T
// Thread (cp = t>>2, kq = t&3) owns hidden unit cp; every reduction is split over the quad's four lanes (k-quarters) and
// combined with quad_sum.  Inputs {r,u,c,dout,h_prev} arrive by LDS-DMA in chunks of CH steps (wave 7).
__global__ __launch_bounds__(NTG, 2) void bigru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                           const float* __restrict__ ruc, BiGruBwdWeights w,
                                                           const float* __restrict__ h0, float* __restrict__ dxg,
                                                           float* __restrict__ rh_out, float* __restrict__ dh0, int B,
                                                           int T) {
  const int b = blockIdx.x, d = blockIdx.y, t_ = threadIdx.x;
  const int cp = t_ >> 2, kq = t_ & 3, lane = t_ & 63, wv = t_ >> 6;
  // double buffered by step parity: the next step's writes never race with this step's reads, so a step needs only the two
  // barriers its own dependences require
  __shared__ __attribute__((aligned(16))) float dcp_s2[2][HP];
  __shared__ __attribute__((aligned(16))) float dgp_s2[2][H2P];
  __shared__ __attribute__((aligned(16))) float in_s[2][CH][5 * H];   // [r | u | c | dout | h_prev]

  // wchT (128 [c], 128 [k]): d(rh)[cp] = sum_c dcp[c] * wchT[c][cp]      -> this lane: c in [32kq, 32kq+32)
  // wghT (256 [j], 128 [k]): dh[cp]   += sum_j dgp[j] * wghT[j][cp]      -> this lane: j in [64kq, 64kq+64)
  f2 wc_r[H / 8], wg_r[H / 4];   // row pairs, for v_pk_fma_f32
  {
    const float* p = w.wchT[d] + (int64_t)(kq * (H / 4)) * H + cp;
#pragma unroll
    for (int i = 0; i < H / 8; ++i) wc_r[i] = f2{p[(int64_t)(2 * i) * H], p[(int64_t)(2 * i + 1) * H]};
    const float* q = w.wghT[d] + (int64_t)(kq * (H / 2)) * H + cp;
#pragma unroll
    for (int i = 0; i < H / 4; ++i) wg_r[i] = f2{q[(int64_t)(2 * i) * H], q[(int64_t)(2 * i + 1) * H]};
  }

  const int64_t row0 = (int64_t)b * T;
  // forward order for d=0 is t = 0..T-1, so backward visits T-1..0; for d=1 forward is T-1..0, backward 0..T-1.
  const int tstart = d == 0 ? T - 1 : 0, tstep = d == 0 ? -1 : 1;
  float dh = 0.f;  // carried gradient of unit cp (kept by every lane of the quad)

  auto dma_chunk = [&](int c) {   // wave i stages step i of chunk c
    const int s = c * CH + wv;
    if (s < T) {
      const int t = tstart + s * tstep;
      float* dst = &in_s[c & 1][wv][0];
      const float* rp = ruc + (row0 + t) * (6 * H) + d * 3 * H + lane;
#pragma unroll
      for (int q = 0; q < 6; ++q) dma64(rp + 64 * q, dst + 64 * q);                    // r | u | c
      const float* dp = dout + (row0 + t) * (2 * H) + d * H + lane;
      dma64(dp, dst + 3 * H);
      dma64(dp + 64, dst + 3 * H + 64);
      if (s + 1 < T) {
        const float* hp = out + (row0 + t + tstep) * (2 * H) + d * H + lane;
        dma64(hp, dst + 4 * H);
        dma64(hp + 64, dst + 4 * H + 64);
      } else if (h0) {
        dma64(h0 + (int64_t)b * H + lane, dst + 4 * H);
        dma64(h0 + (int64_t)b * H + 64 + lane, dst + 4 * H + 64);
      } else {
        dst[4 * H + lane] = 0.f;
        dst[4 * H + 64 + lane] = 0.f;
      }
    }
  };
  dma_chunk(0);
  wait_vm0();
  lds_barrier();

  for (int s = 0, t = tstart; s < T; ++s, t += tstep) {
    const int c = s / CH, i = s - c * CH;
    if (i == 0 && (c + 1) * CH < T) dma_chunk(c + 1);
    float* dcp_s = dcp_s2[s & 1];
    float* dgp_s = dgp_s2[s & 1];
    const float* in = &in_s[c & 1][i][0];
    const float r = in[cp], u = in[H + cp], cc = in[2 * H + cp], hp = in[4 * H + cp];
    const float dht = dh + in[3 * H + cp];
    const float du = dht * (hp - cc);
    const float dc = dht * (1.f - u);
    const float dcp = dc * (1.f - cc * cc);
    const float dup = du * u * (1.f - u);
    if (kq == 0) {
      dcp_s[pad32(cp)] = dcp;
      dgp_s[pad64(H + cp)] = dup;
      float* xo = dxg + (row0 + t) * (6 * H) + d * 3 * H;
      xo[2 * H + cp] = dcp;
      xo[H + cp] = dup;
      rh_out[(row0 + t) * (2 * H) + d * H + cp] = r * hp;
    }
    lds_barrier();
    // d(rh)[cp]
    f2 p0 = {0.f, 0.f}, p1 = p0;
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) {
      const float4 v = reinterpret_cast<const float4*>(dcp_s + kq * (H / 4 + 4))[i4];
      p0 = pk_fma(f2{v.x, v.y}, wc_r[2 * i4], p0);
      p1 = pk_fma(f2{v.z, v.w}, wc_r[2 * i4 + 1], p1);
    }
    const float drh = quad_sum(hsum(p0 + p1));
    const float drp = drh * hp * r * (1.f - r);
    if (kq == 0) {
      dgp_s[pad64(cp)] = drp;
      dxg[(row0 + t) * (6 * H) + d * 3 * H + cp] = drp;
    }
    lds_barrier();
    f2 q0 = {0.f, 0.f}, q1 = q0;
#pragma unroll
    for (int i4 = 0; i4 < H / 8; ++i4) {
      const float4 v = reinterpret_cast<const float4*>(dgp_s + kq * (H / 2 + 4))[i4];
      q0 = pk_fma(f2{v.x, v.y}, wg_r[2 * i4], q0);
      q1 = pk_fma(f2{v.z, v.w}, wg_r[2 * i4 + 1], q1);
    }
    dh = dht * u + drh * r + quad_sum(hsum(q0 + q1));
    if (i == CH - 1) {   // chunk boundary: the next chunk has landed (one store-latency wait per CH steps) and is published
      wait_vm0();
      lds_barrier();
    }
  }
  if (dh0 && kq == 0) dh0[((int64_t)d * B + b) * H + cp] = dh;   // gradient w.r.t. the initial state
}

}  // namespace

int launch_bigru_fwd(const float* xg, const BiGruWeights& w, const float* h0, float* out, float* ruc, int B, int T,
                     hipStream_t s) {
  TACO_REQUIRE(B > 0 && T > 0, "bigru_fwd: bad dims");
  long long* trace = nullptr;
  const int pslot = taco_prof_begin(3, s);
  hipLaunchKernelGGL(bigru_fwd_kernel, dim3(B, 2), dim3(NTG), 0, s, xg, w, h0, out, ruc, B, T, trace);
  taco_prof_end(3, pslot, s, 2.0 * B * T * 2 * (kCb * 2 * kCb + kCb * kCb));   // the h-side mat-vecs of both directions
  TACO_LAUNCH_CHECK("bigru_fwd");
  return TACO_OK;
}

int launch_bigru_bwd(const float* dout, const float* out, const float* ruc, const BiGruBwdWeights& w, const float* h0,
                     float* dxg, float* rh, float* dh0, int B, int T, hipStream_t s) {
  TACO_REQUIRE(B > 0 && T > 0, "bigru_bwd: bad dims");
  // The recurrence is latency bound and runs on only 2B workgroups.  Independent GEMMs are scheduled beside it on a side
  // stream (model.hip); padding this kernel's LDS request to ~148 KB leaves < 16 KB per CU, less than any GEMM tile
  // needs, so those GEMM workgroups fill the OTHER CUs and never share an issue port with the recurrence.
  static DynSmemOnce once;
  const size_t want = 107 * 1024;
  const size_t pad = ensure_dyn_smem(once, reinterpret_cast<const void*>(bigru_bwd_kernel), want) ? want : (size_t)0;
  // (only while the recurrence leaves CUs free for those GEMMs: with more sequences it needs every CU slot itself)
  const int pslot = taco_prof_begin(3, s);
  hipLaunchKernelGGL(bigru_bwd_kernel, dim3(B, 2), dim3(NTG), 2 * B <= 128 ? pad : 0, s, dout, out, ruc, w, h0, dxg, rh, dh0, B, T);
  taco_prof_end(3, pslot, s, 2.0 * B * T * 2 * (kCb * 2 * kCb + kCb * kCb));
  TACO_LAUNCH_CHECK("bigru_bwd");
  return TACO_OK;
}
