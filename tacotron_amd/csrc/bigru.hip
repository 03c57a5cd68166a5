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
// Vector-memory operations of a wave retire in issue order (gfx9 has one counter for loads and stores), so "at most N
// outstanding" = "everything older than the N youngest has landed".  The chunk waits below name the number of STORES the wave
// has issued since the chunk's DMA loads: the loads are then complete while the stores' write acknowledgements (~1 us behind)
// stay in flight -- waiting vmcnt(0) there cost ~150 cycles per step.
template <int N> __device__ __forceinline__ void wait_vm_older_than() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

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
constexpr int HP = H + 16;         // padded 128-vector
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }      // 8 waves = 2 per SIMD: a lone wave per SIMD only issues ~1 instruction per 4 cycles
constexpr int CH = 8;         // time steps per DMA chunk

// Issue-time stamps of wave 0 for tools/micro/gru_trace.hip (compiled out of the library)
#ifdef TACO_GRU_TRACE
#define GRU_STAMP_DECL long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stp_ = 0
#define GRU_STAMP(i)                                                  \
  do {                                                                \
    __builtin_amdgcn_sched_barrier(0);                                \
    const long long n_ = clock64();                                   \
    st_[i] += n_ - stp_;                                              \
    stp_ = n_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                \
  } while (0)
#define GRU_STAMP_FLUSH(tr)                                                           \
  do {                                                                                \
    if ((tr) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)               \
      for (int i_ = 0; i_ < 8; ++i_) (tr)[i_] = st_[i_];                              \
  } while (0)
#else
#define GRU_STAMP_DECL
#define GRU_STAMP(i)
#define GRU_STAMP_FLUSH(tr)
#endif

// Forward.  grid = (B, 2 directions); block = 512.
//  gates    : thread (cp = t>>2, kq = t&3) owns gate columns {cp, cp+128} (= r_cp and u_cp) over k in [32kq, 32kq+32):
//             64 weights in VGPRs, 8 broadcast ds_read_b128 of h, quad_sum combines the four k-quarters in-register.
//  candidate: thread (cc = t>>2, kq) owns candidate column cc over the same k-quarter (32 weights); lane kq==0 of each
//             quad finishes c, h' and issues the stores.  Two workgroup barriers per step.
//  inputs   : the hoisted x-projections arrive by LDS-DMA in chunks of CH steps, issued by wave 7 one chunk ahead.
__global__ __launch_bounds__(NTG, 1) void bigru_fwd_kernel(const float* __restrict__ xg, BiGruWeights w,
                                                           const float* __restrict__ h0, float* __restrict__ out,
                                                           float* __restrict__ ruc, int B, int T, long long* trace) {
  (void)trace;
  GRU_STAMP_DECL;
  const int b = blockIdx.x, d = blockIdx.y, t_ = threadIdx.x;
  const int cp = t_ >> 2, kq = t_ & 3, lane = t_ & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t_ >> 6);   // wave-uniform: the DMA addresses below stay on the scalar unit
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
  const float act_g = kq == 1 ? -1.4426950408889634f : 2.f * 1.4426950408889634f, act_a = kq == 1 ? 0.f : 1.f, act_b = kq == 1 ? 1.f : -2.f;
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
  GRU_STAMP(7);   // (prologue)

  for (int s = 0, t = tstart; s < T; ++s, t += tstep) {
    const int c = s / CH, i = s - c * CH;
    if (i == 0 && (c + 1) * CH < T) dma_chunk(c + 1);
    const float* xrow = &xgs[c & 1][i][0];
    GRU_STAMP(0);   // [0] barrier 2 of the previous step .. loop top
    // ---- reset gate (the only product the candidate waits for) ----
    float4 hv[H / 16];
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) hv[k4] = reinterpret_cast<const float4*>(hs + kq * (H / 4 + 4))[k4];
    f2 a0 = {0.f, 0.f}, a1 = a0;
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) {
      a0 = pk_fma(f2{hv[k4].x, hv[k4].y}, wr[2 * k4], a0);
      a1 = pk_fma(f2{hv[k4].z, hv[k4].w}, wr[2 * k4 + 1], a1);
    }
    GRU_STAMP(1);   // [1] h reads + reset-gate products issued
    const float rg = sigmoid_fast(quad_sum(hsum(a0 + a1)) + xrow[cp]);
    const float hprev = hs[pad32(cp)];
    if (kq == 0) {
      rhs[pad32(cp)] = rg * hprev;
    }
    GRU_STAMP(2);   // [2] quad sum, sigmoid, r*h written
    lds_barrier();
    GRU_STAMP(3);   // [3] barrier 1
    // ---- candidate; the update gate's products (h is still in registers) fill the wait for r*h ----
    float4 rv[H / 16];
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) rv[k4] = reinterpret_cast<const float4*>(rhs + kq * (H / 4 + 4))[k4];
    f2 b0 = {0.f, 0.f}, b1 = b0;
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) {
      b0 = pk_fma(f2{hv[k4].x, hv[k4].y}, wu[2 * k4], b0);
      b1 = pk_fma(f2{hv[k4].z, hv[k4].w}, wu[2 * k4 + 1], b1);
    }
    const float upre = quad_sum(hsum(b0 + b1)) + xrow[H + cp];
    f2 p0 = {0.f, 0.f}, p1 = p0;
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) {
      p0 = pk_fma(f2{rv[k4].x, rv[k4].y}, wcand[2 * k4], p0);
      p1 = pk_fma(f2{rv[k4].z, rv[k4].w}, wcand[2 * k4 + 1], p1);
    }
    GRU_STAMP(4);   // [4] r*h reads, update-gate and candidate products issued
    const float cpre = quad_sum(hsum(p0 + p1)) + xrow[2 * H + cp];
    // every wave's share of the next chunk has had CH-1 steps to land.  A full vmcnt(0): round 4 waited with a COUNT of the stores
    // younger than the chunk's loads (4 or 1 per step), which measured no faster (profiles/r04_gru_lab.txt) and silently depends
    // on how many store instructions the compiler emits per step -- one merged or dropped store and the next chunk
    // would be read before it lands.
    if (i == CH - 1) wait_vm_older_than<0>();
    // sigmoid(u) and tanh(c) share one exp/rcp sequence: lane 1 of the quad takes u, the others c
    //   sigmoid(x) = 0 + 1 * rcp(1 + exp2(-log2e * x));   tanh(x) = 1 - 2 * rcp(1 + exp2(2 log2e * x))     (sigmoid_fast / tanh_fast)
    const float act = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(act_g * (kq == 1 ? upre : cpre))), act_b, act_a);
    const float ug = dpp_move<0xb1>(0.f, act);   // lane 0 <- lane 1
    if (kq == 0) {
      const float cc = act;
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
    GRU_STAMP(5);   // [5] quad sum, tanh, h' written, stores issued
    lds_barrier();
    GRU_STAMP(6);   // [6] barrier 2
  }
  GRU_STAMP_FLUSH(trace);
}

// Backward recurrence.  grid = (B, 2); block = 512.
// Per step (reverse of the forward order), with h_prev = previous forward state, dh = carried gradient:
//   dht = dh + dout; du = dht*(h_prev - c); dc = dht*(1-u); dcp = dc*(1-c^2); d(rh) = dcp . Wc_h^T
//   dr = d(rh)*h_prev; dgp = [dr*r(1-r), du*u(1-u)]; dh = dht*u + d(rh)*r + dgp . Wg_h^T
// Thread (cp = t>>2, kq = t&3) owns hidden unit cp; every reduction is split over the quad's four lanes (k-quarters) and
// combined with quad_sum.  Inputs {r,u,c,dout,h_prev} arrive by LDS-DMA in chunks of CH steps (wave 7).
__global__ __launch_bounds__(NTG, 1) void bigru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                           const float* __restrict__ ruc, BiGruBwdWeights w,
                                                           const float* __restrict__ h0, float* __restrict__ dxg,
                                                           float* __restrict__ rh_out, float* __restrict__ dh0, int B,
                                                           int T) {
  const int b = blockIdx.x, d = blockIdx.y, t_ = threadIdx.x;
  const int cp = t_ >> 2, kq = t_ & 3, lane = t_ & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t_ >> 6);   // wave-uniform: the DMA addresses below stay on the scalar unit
  // double buffered by step parity: the next step's writes never race with this step's reads, so a step needs only the two
  // barriers its own dependences require
  __shared__ __attribute__((aligned(16))) float dcp_s2[2][HP];
  __shared__ __attribute__((aligned(16))) float dgr_s2[2][HP];   // dgp, reset half   (published after d(rh))
  __shared__ __attribute__((aligned(16))) float dgu_s2[2][HP];   // dgp, update half  (known before d(rh))
  __shared__ __attribute__((aligned(16))) float in_s[2][CH][5 * H];   // [r | u | c | dout | h_prev]

  // wchT (128 [c], 128 [k]): d(rh)[cp] = sum_c dcp[c] * wchT[c][cp]      -> this lane: c in [32kq, 32kq+32)
  // wghT (256 [j], 128 [k]): dh[cp]   += sum_j dgp[j] * wghT[j][cp]      -> this lane: j in [32kq, 32kq+32) (reset half)
  //                                                                        and [128+32kq, 128+32kq+32) (update half)
  f2 wc_r[H / 8], wgr_r[H / 8], wgu_r[H / 8];   // row pairs, for v_pk_fma_f32
  {
    const float* p = w.wchT[d] + (int64_t)(kq * (H / 4)) * H + cp;
#pragma unroll
    for (int i = 0; i < H / 8; ++i) wc_r[i] = f2{p[(int64_t)(2 * i) * H], p[(int64_t)(2 * i + 1) * H]};
    const float* q = w.wghT[d] + (int64_t)(kq * (H / 4)) * H + cp;
#pragma unroll
    for (int i = 0; i < H / 8; ++i) {
      wgr_r[i] = f2{q[(int64_t)(2 * i) * H], q[(int64_t)(2 * i + 1) * H]};
      wgu_r[i] = f2{q[(int64_t)(H + 2 * i) * H], q[(int64_t)(H + 2 * i + 1) * H]};
    }
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
    float* dgr_s = dgr_s2[s & 1];
    float* dgu_s = dgu_s2[s & 1];
    const float* in = &in_s[c & 1][i][0];
    const float r = in[cp], u = in[H + cp], cc = in[2 * H + cp], hp = in[4 * H + cp];
    const float dht = dh + in[3 * H + cp];
    const float du = dht * (hp - cc);
    const float dc = dht * (1.f - u);
    const float dcp = dc * (1.f - cc * cc);
    const float dup = du * u * (1.f - u);
    if (kq == 0) {
      dcp_s[pad32(cp)] = dcp;
      dgu_s[pad32(cp)] = dup;
      float* xo = dxg + (row0 + t) * (6 * H) + d * 3 * H;
      xo[2 * H + cp] = dcp;
      xo[H + cp] = dup;
      rh_out[(row0 + t) * (2 * H) + d * H + cp] = r * hp;
    }
    lds_barrier();
    // d(rh)[cp]; the update half of dgp is fetched beside dcp and multiplied while the reset half is still in flight
    float4 dv[H / 16], uv[H / 16];
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) dv[i4] = reinterpret_cast<const float4*>(dcp_s + kq * (H / 4 + 4))[i4];
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) uv[i4] = reinterpret_cast<const float4*>(dgu_s + kq * (H / 4 + 4))[i4];
    f2 p0 = {0.f, 0.f}, p1 = p0;
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) {
      p0 = pk_fma(f2{dv[i4].x, dv[i4].y}, wc_r[2 * i4], p0);
      p1 = pk_fma(f2{dv[i4].z, dv[i4].w}, wc_r[2 * i4 + 1], p1);
    }
    const float drh = quad_sum(hsum(p0 + p1));
    const float drp = drh * hp * r * (1.f - r);
    if (kq == 0) {
      dgr_s[pad32(cp)] = drp;
      dxg[(row0 + t) * (6 * H) + d * 3 * H + cp] = drp;
    }
    lds_barrier();
    float4 gv[H / 16];
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) gv[i4] = reinterpret_cast<const float4*>(dgr_s + kq * (H / 4 + 4))[i4];
    f2 q0 = {0.f, 0.f}, q1 = q0;
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) {
      q0 = pk_fma(f2{uv[i4].x, uv[i4].y}, wgu_r[2 * i4], q0);
      q1 = pk_fma(f2{uv[i4].z, uv[i4].w}, wgu_r[2 * i4 + 1], q1);
    }
#pragma unroll
    for (int i4 = 0; i4 < H / 16; ++i4) {
      q0 = pk_fma(f2{gv[i4].x, gv[i4].y}, wgr_r[2 * i4], q0);
      q1 = pk_fma(f2{gv[i4].z, gv[i4].w}, wgr_r[2 * i4 + 1], q1);
    }
    dh = dht * u + drh * r + quad_sum(hsum(q0 + q1));
    if (i == CH - 1) {   // chunk boundary: the next chunk has landed (vmcnt(0), not a store count: see the forward kernel) and is published
      wait_vm_older_than<0>();
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
  TACO_KLAUNCH(bigru_fwd_kernel, dim3(B, 2), dim3(NTG), 0, s, xg, w, h0, out, ruc, B, T, trace);
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
  TACO_KLAUNCH(bigru_bwd_kernel, dim3(B, 2), dim3(NTG), 2 * B <= 128 ? pad : 0, s, dout, out, ruc, w, h0, dxg, rh, dh0, B, T);
  taco_prof_end(3, pslot, s, 2.0 * B * T * 2 * (kCb * 2 * kCb + kCb * kCb));
  TACO_LAUNCH_CHECK("bigru_bwd");
  return TACO_OK;
}
