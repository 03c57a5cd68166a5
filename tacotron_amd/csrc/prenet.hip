// prenet.hip -- the encoder pre_net (tacotron.py:38-44: two dense + ReLU + dropout layers, 256 -> 256 -> 128) and its
// activation-gradient chain as ONE launch each.
//
// Same shape of kernel as the highway stacks (highway.hip): a workgroup keeps a 32-row tile in LDS (feature-major, pitch 33)
// and walks both layers; the MFMA B operand -- one weight per lane, W[k = 2j + (lane >> 5)][n] -- is fetched by coalesced
// global_load_dwords straight into registers, one 32-k chunk ahead of the MFMAs that use it (all workgroups read the same
// weights: L2 hits), so nothing but the activations passes through LDS and a layer's k-loop has no barrier.  Four waves; a wave
// owns N / 4 consecutive output columns (one or two 32-wide sub-tiles).  As two conv_gemm launches the forward pair was 56 us
// of the critical path at M = 6400 (100-400 tiles of a kernel that wants thousands); backward, four launches.
//
//   forward :  p1 = drop1(relu(x W1 + b1)),  p2 = drop2(relu(p1 W2 + b2))                      (p1, p2 kept for the backward pass)
//   backward:  dz2 = d p2 * [p2 > 0] * drop2;  dz1 = (dz2 W2^T) * [p1 > 0] * drop1;  dx = dz1 W1^T   (dz2, dz1 feed the
//              weight-gradient GEMMs; W^T are the transposed copies prepare_transposes builds)
#include "common.h"
#include "kernels.h"

namespace {

#ifdef TACO_PN_TRACE   // probe build: shader-clock stamps of workgroup 0 at the stage boundaries
#define PN_STAMP(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[i] = clock64(); } while (0)
#else
#define PN_STAMP(i) do { } while (0)
#endif
#ifdef TACO_PN_P_NOW   // timing probe (tools/ab_build.sh): no weight loads at all -- results are garbage
#define PN_W(x) (0.001f * (float)(j + s))
#else
#define PN_W(x) (x)
#endif

constexpr int PB = 32;          // rows per workgroup
constexpr int PPAD = PB + 1;    // activation pitch, feature-major
constexpr int PCK = 16;         // k-pairs per chunk (32 k-values)

template <int K1, int N1, int N2, bool BWD>
__global__ __launch_bounds__(256, 2) void mlp2_kernel(PrenetArgs a) {
  constexpr int S1 = N1 / 128, S2 = N2 / 128;
  static_assert(S1 >= 1 && S1 <= 2 && S2 >= 1 && S2 <= 2 && K1 % 32 == 0 && N1 % 32 == 0, "pre_net widths");
  extern __shared__ __attribute__((aligned(16))) float pn_smem[];
  float (*aT)[PPAD] = reinterpret_cast<float (*)[PPAD]>(pn_smem);               // [K1][33] stage-1 input
  float (*bT)[PPAD] = reinterpret_cast<float (*)[PPAD]>(pn_smem + K1 * PPAD);   // [N1][33] stage-2 input
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 5, li = lane & 31;
  const int m0 = blockIdx.x * PB;

  PN_STAMP(0);
  float wr[2][2][PCK];
  // first chunk of stage 1 under the input tile's load
  {
    const float* src = a.w1 + (int64_t)lk * N1 + wave * 32 * S1 + li;
#pragma unroll
    for (int j = 0; j < PCK; ++j)
#pragma unroll
      for (int s = 0; s < S1; ++s) wr[0][s][j] = PN_W(src[(int64_t)2 * j * N1 + 32 * s]);
  }
  // ---- stage 0: input tile -> aT (backward: through the ReLU / dropout of layer 2, and out to dz2) ----
  {
    const int row = tid >> 3, kq = tid & 7;
    const int m = m0 + row;
#pragma unroll
    for (int i = 0; i < K1 / 32; ++i) {
      const int k = (kq + 8 * i) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M) {
        v = *reinterpret_cast<const float4*>(a.x + (int64_t)m * K1 + k);
        if (BWD) {
          const float4 y = *reinterpret_cast<const float4*>(a.y2_in + (int64_t)m * K1 + k);
          uint32_t kp = 0x01010101u;
          if (a.keep2) kp = *reinterpret_cast<const uint32_t*>(a.keep2 + (int64_t)m * K1 + k);
          const float sc = a.keep2 ? 2.f : 1.f;
          v.x = (y.x > 0.f && (kp & 0xffu)) ? v.x * sc : 0.f;
          v.y = (y.y > 0.f && (kp & 0xff00u)) ? v.y * sc : 0.f;
          v.z = (y.z > 0.f && (kp & 0xff0000u)) ? v.z * sc : 0.f;
          v.w = (y.w > 0.f && (kp & 0xff000000u)) ? v.w * sc : 0.f;
          *reinterpret_cast<float4*>(a.x_out + (int64_t)m * K1 + k) = v;
        }
      }
      aT[k + 0][row] = v.x;
      aT[k + 1][row] = v.y;
      aT[k + 2][row] = v.z;
      aT[k + 3][row] = v.w;
    }
  }
  lds_barrier();
  PN_STAMP(1);

  // ---- stage 1 ----
  f32x16 acc[2][2];
  {
    // (chunk 0 is already in set 0)
    constexpr int NC = K1 / (2 * PCK);
    auto load_chunk = [&](int set, int c) {
      const float* src = a.w1 + (int64_t)(2 * c * PCK + lk) * N1 + wave * 32 * S1 + li;
#pragma unroll
      for (int j = 0; j < PCK; ++j)
#pragma unroll
        for (int s = 0; s < S1; ++s) wr[set][s][j] = PN_W(src[(int64_t)2 * j * N1 + 32 * s]);
    };
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s][0][e] = acc[s][1][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        load_chunk((c + 1) & 1, c + 1);
      } else {   // the first chunk of stage 2 flies under the last chunk of stage 1 and its epilogue
        const float* src = a.w2 + (int64_t)lk * N2 + wave * 32 * S2 + li;
#pragma unroll
        for (int j = 0; j < PCK; ++j)
#pragma unroll
          for (int s = 0; s < S2; ++s) wr[(c + 1) & 1][s][j] = PN_W(src[(int64_t)2 * j * N2 + 32 * s]);
      }
      __builtin_amdgcn_sched_barrier(0);
      float av[PCK];
#pragma unroll
      for (int j = 0; j < PCK; ++j) av[j] = aT[2 * (c * PCK + j) + lk][li];
#pragma unroll
      for (int j = 0; j < PCK; ++j)
#pragma unroll
        for (int s = 0; s < S1; ++s)
          acc[s][j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wr[c & 1][s][j], acc[s][j & 1], 0, 0, 0);
    }
  }
  PN_STAMP(2);
  // epilogue 1: C layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int s = 0; s < S1; ++s) {
    const int n = wave * 32 * S1 + 32 * s + li;
    const float b = (!BWD && a.b1) ? a.b1[n] : 0.f;
    // every mask operand of the 16 elements is loaded BEFORE the first store: the keep mask is a byte array (it may alias
    // anything as far as the compiler knows), so a load behind a store would wait for that store -- 32 serial round trips
    // (and they stay RAW, one 32-bit register each: a byte array would be packed four to a register, i.e. arithmetic -- and a
    //  wait -- behind every single load)
    float yin[16];
    unsigned kp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) yin[e] = 1.f, kp[e] = 1u;
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        yin[e] = a.y1_in[(int64_t)(m < a.M ? m : a.M - 1) * N1 + n];
      }
    }
    if (a.keep1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        kp[e] = a.keep1[(int64_t)(m < a.M ? m : a.M - 1) * N1 + n];
      }
    }
    // every loaded value is consumed before the first store as well: a use of a loaded register BEHIND a conditional store gets
    // `s_waitcnt vmcnt(0)` (the number of younger stores is path dependent), i.e. a wait for that store's acknowledgement --
    // one per element, 17 us of a 45 us workgroup when this loop was written the natural way
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
      float v = acc[s][0][e] + acc[s][1][e] + b;
      if (BWD) v = yin[e] > 0.f ? v : 0.f;
      else v = fmaxf(v, 0.f);
      v = kp[e] ? (a.keep1 ? v * 2.f : v) : 0.f;
      acc[s][0][e] = m < a.M ? v : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lk;
      const int m = m0 + r;
      if (m < a.M) a.y1[(int64_t)m * N1 + n] = acc[s][0][e];
      bT[n][r] = acc[s][0][e];
    }
  }
  PN_STAMP(3);
  lds_barrier();
  PN_STAMP(4);

  // ---- stage 2 ----
  constexpr int NC1 = K1 / (2 * PCK);
  {
    constexpr int NC = N1 / (2 * PCK);
    auto load_chunk = [&](int set, int c) {
      const float* src = a.w2 + (int64_t)(2 * c * PCK + lk) * N2 + wave * 32 * S2 + li;
#pragma unroll
      for (int j = 0; j < PCK; ++j)
#pragma unroll
        for (int s = 0; s < S2; ++s) wr[set][s][j] = PN_W(src[(int64_t)2 * j * N2 + 32 * s]);
    };
#pragma unroll
    for (int s = 0; s < S2; ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s][0][e] = acc[s][1][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int set = (NC1 + c) & 1;   // chunk 0 of this stage was loaded into set NC1 & 1 above
      if (c + 1 < NC) load_chunk(set ^ 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      float av[PCK];
#pragma unroll
      for (int j = 0; j < PCK; ++j) av[j] = bT[2 * (c * PCK + j) + lk][li];
#pragma unroll
      for (int j = 0; j < PCK; ++j)
#pragma unroll
        for (int s = 0; s < S2; ++s)
          acc[s][j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wr[set][s][j], acc[s][j & 1], 0, 0, 0);
    }
  }
  PN_STAMP(5);
#pragma unroll
  for (int s = 0; s < S2; ++s) {
    const int n = wave * 32 * S2 + 32 * s + li;
    const float b = (!BWD && a.b2) ? a.b2[n] : 0.f;
    unsigned kp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) kp[e] = 1u;
    if (!BWD && a.keep2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        kp[e] = a.keep2[(int64_t)(m < a.M ? m : a.M - 1) * N2 + n];
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[s][0][e] + acc[s][1][e] + b;
      if (!BWD) {
        v = fmaxf(v, 0.f);
        v = kp[e] ? (a.keep2 ? v * 2.f : v) : 0.f;
      }
      acc[s][0][e] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
      if (m < a.M) a.y2[(int64_t)m * N2 + n] = acc[s][0][e];
    }
  }
  PN_STAMP(6);
}

template <int K1, int N1, int N2, bool BWD>
int launch_mlp2(const PrenetArgs& a, hipStream_t s, const char* what) {
  constexpr size_t smem = sizeof(float) * (size_t)(K1 + N1) * PPAD;
  static DynSmemOnce once;
  TACO_REQUIRE(ensure_dyn_smem(once, reinterpret_cast<const void*>(mlp2_kernel<K1, N1, N2, BWD>), smem),
               "%s: cannot reserve %zu bytes of LDS", what, smem);
  const int pslot = taco_prof_begin(2, s);
  taco_prof_label(2, pslot, "%s M=%d", what, a.M);
  TACO_KLAUNCH((mlp2_kernel<K1, N1, N2, BWD>), dim3(cdiv(a.M, PB)), dim3(256), smem, s, a);
  taco_prof_end(2, pslot, s, 2.0 * a.M * ((double)K1 * N1 + (double)N1 * N2));
  return TACO_OK;
}

}  // namespace

int launch_prenet_fwd(const PrenetArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.x && a.w1 && a.w2 && a.y1 && a.y2, "prenet_fwd: bad arguments");
  TACO_TRY((launch_mlp2<kEmbed, kPre1, kPre2, false>(a, s, "prenet_fwd")));
  TACO_LAUNCH_CHECK("prenet_fwd");
  return TACO_OK;
}

int launch_prenet_bwd(const PrenetArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.M > 0 && a.x && a.x_out && a.w1 && a.w2 && a.y1 && a.y2 && a.y1_in && a.y2_in, "prenet_bwd: bad arguments");
  TACO_TRY((launch_mlp2<kPre2, kPre1, kEmbed, true>(a, s, "prenet_bwd")));
  TACO_LAUNCH_CHECK("prenet_bwd");
  return TACO_OK;
}
