// decoder.hip -- persistent attention-decoder kernels (tacotron.py:46-105 create_decoder + :134-138 dynamic_decode,
// with the TF-r1.2 AttentionWrapper / BahdanauAttention / GRUCell / projection-wrapper / helper semantics restated
// in SURVEY.md §8a rows a11-a15).
//
// Design (round 1): ONE workgroup of 512 threads owns ONE batch row for ALL Td steps -- a single launch replaces
// the reference's 180-iteration tf.while_loop (~13k op launches).  All recurrent state (3 GRU states, attention
// vector, previous frame, alignments) stays in LDS across steps; per step the row streams the 1.59 M decoder weights
// (6.35 MB, L2/MALL-resident and shared by all rows) through a split-K mat-vec (float4 loads, 8 waves in flight) and
// its own keys/values (2 x Tt x 256 floats).  There is no inter-workgroup communication, hence no dispatch-order or
// XCD-placement assumption.  Attention energies/softmax/context are a fused wave-reduction phase (one wave per
// memory row, __shfl_xor reductions).
//
// The backward kernel walks the steps in reverse with the same structure on pre-transposed weights and emits the
// per-step pre-activation gradients ("gstash"); all weight gradients are then dense MFMA GEMMs over B*Td rows.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NT = 512;
constexpr int kPartFloats = NT * 4;

// y[n] = sum_k x[k] * W[k*ldw + n] for n < N; `fin(n, y)` is invoked by the owning thread(s) for every n.
// x lives in LDS and must be readable up to K rounded up to 4.  N % 4 == 0.  Contains ONE __syncthreads(); the
// caller must __syncthreads() after its epilogue before `part`/x are reused.
template <class Fin>
__device__ __forceinline__ void matvec(const float* __restrict__ W, int ldw, int K, int N, const float* x, float* part,
                                       Fin fin) {
  const int tid = threadIdx.x;
  const int N4 = N >> 2;
  const int KG = NT / N4;
  const int kg = tid / N4, c4 = tid - kg * N4;
  if (kg < KG) {
    const int Kc = (((K + KG - 1) / KG) + 3) & ~3;
    const int k0 = kg * Kc;
    const int k1 = min(K, k0 + Kc);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = W + (int64_t)k0 * ldw + c4 * 4;
    int k = k0;
#pragma unroll 2
    for (; k + 3 < k1; k += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(x + k);
      const float4 w0 = *reinterpret_cast<const float4*>(wp);
      const float4 w1 = *reinterpret_cast<const float4*>(wp + ldw);
      const float4 w2 = *reinterpret_cast<const float4*>(wp + 2 * (int64_t)ldw);
      const float4 w3 = *reinterpret_cast<const float4*>(wp + 3 * (int64_t)ldw);
      wp += 4 * (int64_t)ldw;
      acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
      acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
      acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
      acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
    }
    for (; k < k1; ++k) {
      const float xs = x[k];
      const float4 w0 = *reinterpret_cast<const float4*>(wp);
      wp += ldw;
      acc.x = fmaf(xs, w0.x, acc.x); acc.y = fmaf(xs, w0.y, acc.y); acc.z = fmaf(xs, w0.z, acc.z); acc.w = fmaf(xs, w0.w, acc.w);
    }
    *reinterpret_cast<float4*>(part + kg * N + c4 * 4) = acc;
  }
  __syncthreads();
  for (int n = tid; n < N; n += NT) {
    float y = 0.f;
    for (int g = 0; g < KG; ++g) y += part[g * N + n];
    fin(n, y);
  }
}

struct DecSmem {
  float* part;   // kPartFloats
  float* fr;     // 80   pre-net input frame
  float* p1;     // 256
  float* xin;    // 384  [p2 ; attention]
  float* xs;     // 256  in-proj output (residual)
  float* cat;    // 3*512 [layer input ; h_l]
  float* catc;   // 512  [layer input ; r*h_l]
  float* us;     // 256
  float* ys;     // 256
  float* octx;   // 656  [cell_output (80r) ; context (256)]
  float* qs;     // 256
  float* es;     // TtP energies
  float* als;    // TtP alignments
};

__device__ __forceinline__ DecSmem carve(float* base, int TtP) {
  DecSmem s;
  float* p = base;
  s.part = p; p += kPartFloats;
  s.fr = p; p += 80;
  s.p1 = p; p += 256;
  s.xin = p; p += 384;
  s.xs = p; p += 256;
  s.cat = p; p += 3 * 512;
  s.catc = p; p += 512;
  s.us = p; p += 256;
  s.ys = p; p += 256;
  s.octx = p; p += 656;
  s.qs = p; p += 256;
  s.es = p; p += TtP;
  s.als = p; p += TtP;
  return s;
}
constexpr int kFwdSmemFixed = kPartFloats + 80 + 256 + 384 + 256 + 3 * 512 + 512 + 256 + 256 + 656 + 256;

__global__ __launch_bounds__(NT) void decoder_fwd_kernel(DecFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int B = a.B, Tt = a.Tt, Td = a.Td, r = a.r;
  const int R80 = kMel * r;
  const int TtP = (Tt + 3) & ~3;
  DecSmem S = carve(smem, TtP);
  const DecWeights& w = a.w;

  int len = a.text_length[b];
  len = len < 1 ? 1 : (len > Tt ? Tt : len);
  const float* keys = a.keys + (int64_t)b * Tt * kAtt;
  const float* values = a.values + (int64_t)b * Tt * kAtt;

  // zero state (AttentionWrapper.zero_state, tacotron.py:94)
  for (int i = tid; i < 3 * 512; i += NT) S.cat[i] = 0.f;
  for (int i = tid; i < 384; i += NT) S.xin[i] = 0.f;
  for (int i = tid; i < TtP; i += NT) { S.als[i] = 0.f; S.es[i] = 0.f; }
  for (int i = tid; i < 656; i += NT) S.octx[i] = 0.f;
  if (tid < kMel) S.fr[tid] = a.mel ? a.mel[((int64_t)b * Td) * R80 + kMel * (r - 1) + tid] : 0.f;
  // per-thread constants
  const float4 v4 = reinterpret_cast<const float4*>(w.att_v)[lane];
  __syncthreads();

  for (int t = 0; t < Td; ++t) {
    const int64_t bt = (int64_t)b * Td + t;
    float* st = a.stash ? a.stash + bt * kStRec : nullptr;
    if (a.prein && tid < kMel) a.prein[bt * kMel + tid] = S.fr[tid];

    // ---- pre_net (tacotron.py:38-44, 64-71) ----
    matvec(w.pre_w1, kPre1, kMel, kPre1, S.fr, S.part, [&](int n, float y) {
      y = fmaxf(y + w.pre_b1[n], 0.f);
      if (a.keep1) y = a.keep1[bt * kPre1 + n] ? 2.f * y : 0.f;
      S.p1[n] = y;
      if (st) st[kStP1 + n] = y;
    });
    __syncthreads();
    matvec(w.pre_w2, kPre2, kPre1, kPre2, S.p1, S.part, [&](int n, float y) {
      y = fmaxf(y + w.pre_b2[n], 0.f);
      if (a.keep2) y = a.keep2[bt * kPre2 + n] ? 2.f * y : 0.f;
      S.xin[n] = y;
      if (st) st[kStP2 + n] = y;
    });
    __syncthreads();
    // ---- InputProjectionWrapper: x = [pre_net ; attention] Wi + bi (tacotron.py:56-59) ----
    matvec(w.in_w, kDec, kPre2 + kAtt, kDec, S.xin, S.part, [&](int n, float y) {
      y += w.in_b[n];
      S.xs[n] = y;
      S.cat[n] = y;
      S.catc[n] = y;
      if (st) st[kStX + n] = y;
    });
    __syncthreads();
    // ---- MultiRNNCell[GRUCell(256) x3] inside ONE ResidualWrapper (tacotron.py:54-58) ----
    for (int l = 0; l < 3; ++l) {
      float* cl = S.cat + l * 512;
      matvec(w.gw[l], 2 * kDec, 2 * kDec, 2 * kDec, cl, S.part, [&](int n, float y) {
        const float g = sigmoid_f(y + w.gb[l][n]);
        if (n < kDec) {
          const float rh = g * cl[kDec + n];
          S.catc[kDec + n] = rh;
          if (st) { st[kStR + l * kDec + n] = g; st[kStRH + l * kDec + n] = rh; }
        } else {
          S.us[n - kDec] = g;
          if (st) st[kStU + l * kDec + n - kDec] = g;
        }
      });
      __syncthreads();
      matvec(w.cw[l], kDec, 2 * kDec, kDec, S.catc, S.part, [&](int n, float y) {
        const float c = tanh_f(y + w.cb[l][n]);
        const float u = S.us[n];
        const float hn = u * cl[kDec + n] + (1.f - u) * c;
        cl[kDec + n] = hn;
        if (l < 2) {
          S.cat[(l + 1) * 512 + n] = hn;
          S.catc[n] = hn;
        } else {
          S.ys[n] = S.xs[n] + hn;
        }
        if (st) {
          st[kStC + l * kDec + n] = c;
          st[kStH + l * kDec + n] = hn;
          if (l == 2) st[kStY + n] = S.xs[n] + hn;
        }
      });
      __syncthreads();
    }
    // ---- OutputProjectionWrapper: cell_output = (x + h3) Wo + bo (tacotron.py:54-60) ----
    matvec(w.out_w, R80, kDec, R80, S.ys, S.part, [&](int n, float y) {
      y += w.out_b[n];
      S.octx[n] = y;
      a.out[bt * R80 + n] = y;
    });
    __syncthreads();
    // ---- BahdanauAttention: query layer (no bias) ----
    matvec(w.q_w, kAtt, R80, kAtt, S.octx, S.part, [&](int n, float y) {
      S.qs[n] = y;
      if (st) st[kStQ + n] = y;
    });
    __syncthreads();
    // ---- energies e[s] = sum_u v_u tanh(keys[s,u] + q_u): one wave per memory row ----
    {
      const float4 q4 = reinterpret_cast<const float4*>(S.qs)[lane];
      for (int s = wave; s < len; s += NT / 64) {
        const float4 k4 = reinterpret_cast<const float4*>(keys + (int64_t)s * kAtt)[lane];
        float e = v4.x * tanh_f(k4.x + q4.x) + v4.y * tanh_f(k4.y + q4.y) + v4.z * tanh_f(k4.z + q4.z) +
                  v4.w * tanh_f(k4.w + q4.w);
        e = wave_sum(e);
        if (lane == 0) S.es[s] = e;
      }
    }
    __syncthreads();
    // ---- masked softmax over s < len (score_mask_value = -inf => alignment 0 past text_length) ----
    {
      float m = -INFINITY;
      for (int s = lane; s < len; s += 64) m = fmaxf(m, S.es[s]);
      m = wave_max(m);
      float z = 0.f;
      for (int s = lane; s < len; s += 64) z += expf(S.es[s] - m);
      z = wave_sum(z);
      const float inv = 1.0f / z;
      for (int s = tid; s < Tt; s += NT) {
        const float al = s < len ? expf(S.es[s] - m) * inv : 0.f;
        S.als[s] = al;
        a.align[bt * Tt + s] = al;
      }
    }
    __syncthreads();
    // ---- context = alignments . values ----
    matvec(values, kAtt, len, kAtt, S.als, S.part, [&](int n, float y) {
      S.octx[R80 + n] = y;
      if (st) st[kStCtx + n] = y;
    });
    __syncthreads();
    // ---- attention = [cell_output ; context] Wa (attention_layer_size=256, no bias; tacotron.py:76) ----
    matvec(w.att_w, kAtt, R80 + kAtt, kAtt, S.octx, S.part, [&](int n, float y) {
      S.xin[kPre2 + n] = y;
      if (st) st[kStAtt + n] = y;
    });
    // ---- helper.next_inputs: TrainingHelper / ScheduledOutputTrainingHelper / InferenceHelper ----
    if (tid < kMel && t + 1 < Td) {
      float nf;
      const bool from_out = (a.mel == nullptr) || (a.sample && a.sample[(int64_t)t * B + b]);
      if (from_out) nf = S.octx[kMel * (r - 1) + tid];
      else nf = a.mel[(bt + 1) * R80 + kMel * (r - 1) + tid];
      S.fr[tid] = nf;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------
struct DecBwdSmem {
  float* part;    // kPartFloats
  float* dh;      // 3*256 carried dL/dh_l
  float* datt;    // 256 carried dL/d attention_{t}
  float* dfr;     // 80  dL/d next step's pre-net input frame
  float* dov;     // 656 [d cell_output (80r) ; d context (256)]
  float* dq;      // 256
  float* dy;      // 256 d(x + h3)
  float* dht;     // 256 total dL/dh_l at this step
  float* dcp;     // 256
  float* dgp;     // 512
  float* dinp;    // 256 gradient into the layer input
  float* dx;      // 256
  float* dxin;    // 384
  float* dp2;     // 128
  float* dp1;     // 256
  float* qs;      // 256
  float* als;     // TtP
  float* des;     // TtP
  float* red;     // 8*256 cross-wave dq reduction
};
constexpr int kBwdSmemFixed =
    kPartFloats + 768 + 256 + 80 + 656 + 256 + 256 + 256 + 256 + 512 + 256 + 256 + 384 + 128 + 256 + 256 + 8 * 256;

__device__ __forceinline__ DecBwdSmem carve_bwd(float* base, int TtP) {
  DecBwdSmem s;
  float* p = base;
  s.part = p; p += kPartFloats;
  s.dh = p; p += 768;
  s.datt = p; p += 256;
  s.dfr = p; p += 80;
  s.dov = p; p += 656;
  s.dq = p; p += 256;
  s.dy = p; p += 256;
  s.dht = p; p += 256;
  s.dcp = p; p += 256;
  s.dgp = p; p += 512;
  s.dinp = p; p += 256;
  s.dx = p; p += 256;
  s.dxin = p; p += 384;
  s.dp2 = p; p += 128;
  s.dp1 = p; p += 256;
  s.qs = p; p += 256;
  s.red = p; p += 8 * 256;
  s.als = p; p += TtP;
  s.des = p; p += TtP;
  return s;
}

__global__ __launch_bounds__(NT) void decoder_bwd_kernel(DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int B = a.B, Tt = a.Tt, Td = a.Td, r = a.r;
  const int R80 = kMel * r;
  const int TtP = (Tt + 3) & ~3;
  DecBwdSmem S = carve_bwd(smem, TtP);
  const DecWeights& w = a.wT;

  int len = a.text_length[b];
  len = len < 1 ? 1 : (len > Tt ? Tt : len);
  const float* keys = a.keys + (int64_t)b * Tt * kAtt;
  const float* values = a.values + (int64_t)b * Tt * kAtt;
  float* dkeys = a.dkeys + (int64_t)b * Tt * kAtt;

  for (int i = tid; i < 768; i += NT) S.dh[i] = 0.f;
  if (tid < 256) S.datt[tid] = 0.f;
  if (tid < 80) S.dfr[tid] = 0.f;
  for (int i = tid; i < TtP; i += NT) { S.als[i] = 0.f; S.des[i] = 0.f; }
  const float4 v4 = reinterpret_cast<const float4*>(a.att_v)[lane];
  float4 dv4 = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  for (int t = Td - 1; t >= 0; --t) {
    const int64_t bt = (int64_t)b * Td + t;
    const float* st = a.stash + bt * kStRec;
    const float* stp = t > 0 ? a.stash + (bt - 1) * kStRec : nullptr;
    float* gs = a.gstash + bt * kGsRec;
    const bool next_from_out = (t + 1 < Td) && a.sample && a.sample[(int64_t)t * B + b];
    const bool this_from_out = (t > 0) && a.sample && a.sample[(int64_t)(t - 1) * B + b];

    // 1. d cell_output: direct (loss + post-net) + sampled next-input path
    for (int n = tid; n < R80; n += NT) {
      float g = a.dout[bt * R80 + n];
      if (next_from_out && n >= kMel * (r - 1)) g += S.dfr[n - kMel * (r - 1)];
      S.dov[n] = g;
    }
    if (tid < kAtt) gs[kGsAtt + tid] = S.datt[tid];
    for (int s = tid; s < Tt; s += NT) S.als[s] = a.align[bt * Tt + s];
    if (tid < kAtt) S.qs[tid] = st[kStQ + tid];
    __syncthreads();
    // 2. attention layer: d[o ; ctx] += datt . Wa^T      (wT.att_w is (256, 80r+256))
    matvec(w.att_w, R80 + kAtt, kAtt, R80 + kAtt, S.datt, S.part, [&](int n, float y) {
      if (n < R80) S.dov[n] += y;
      else {
        S.dov[n] = y;
        gs[kGsCtx + n - R80] = y;
      }
    });
    __syncthreads();
    // 3a. d alignments[s] = values[s] . dctx
    {
      const float4 c4 = reinterpret_cast<const float4*>(S.dov + R80)[lane];
      for (int s = wave; s < len; s += NT / 64) {
        const float4 x4 = reinterpret_cast<const float4*>(values + (int64_t)s * kAtt)[lane];
        float d = x4.x * c4.x + x4.y * c4.y + x4.z * c4.z + x4.w * c4.w;
        d = wave_sum(d);
        if (lane == 0) S.des[s] = d;
      }
    }
    __syncthreads();
    // 3b. softmax backward: de = al * (dal - sum al*dal)   (every wave computes the dot redundantly)
    {
      float dot = 0.f;
      for (int s = lane; s < len; s += 64) dot += S.als[s] * S.des[s];
      dot = wave_sum(dot);
      __syncthreads();
      for (int s = tid; s < len; s += NT) S.des[s] = S.als[s] * (S.des[s] - dot);
    }
    __syncthreads();
    // 3c. energy backward: th = tanh(keys+q); dpre = de*v*(1-th^2); dq += dpre; dkeys += dpre; dv += de*th
    {
      const float4 q4 = reinterpret_cast<const float4*>(S.qs)[lane];
      float4 dq4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = wave; s < len; s += NT / 64) {
        const float de = S.des[s];
        const float4 k4 = reinterpret_cast<const float4*>(keys + (int64_t)s * kAtt)[lane];
        float4 dk = reinterpret_cast<float4*>(dkeys + (int64_t)s * kAtt)[lane];
        const float t0 = tanh_f(k4.x + q4.x), t1 = tanh_f(k4.y + q4.y), t2 = tanh_f(k4.z + q4.z), t3 = tanh_f(k4.w + q4.w);
        const float p0 = de * v4.x * (1.f - t0 * t0), p1 = de * v4.y * (1.f - t1 * t1);
        const float p2 = de * v4.z * (1.f - t2 * t2), p3 = de * v4.w * (1.f - t3 * t3);
        dq4.x += p0; dq4.y += p1; dq4.z += p2; dq4.w += p3;
        dk.x += p0; dk.y += p1; dk.z += p2; dk.w += p3;
        reinterpret_cast<float4*>(dkeys + (int64_t)s * kAtt)[lane] = dk;
        dv4.x += de * t0; dv4.y += de * t1; dv4.z += de * t2; dv4.w += de * t3;
      }
      reinterpret_cast<float4*>(S.red + wave * 256)[lane] = dq4;
    }
    __syncthreads();
    if (tid < kAtt) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < NT / 64; ++i) d += S.red[i * 256 + tid];
      S.dq[tid] = d;
      gs[kGsQ + tid] = d;
    }
    __syncthreads();
    // 4. query layer: do += dq . Wq^T   (wT.q_w is (256, 80r))
    matvec(w.q_w, R80, kAtt, R80, S.dq, S.part, [&](int n, float y) {
      const float g = S.dov[n] + y;
      S.dov[n] = g;
      gs[kGsO + n] = g;
    });
    __syncthreads();
    // 5. output projection: dy = do . Wo^T   (wT.out_w is (80r, 256))
    matvec(w.out_w, kDec, R80, kDec, S.dov, S.part, [&](int n, float y) {
      S.dy[n] = y;
      S.dht[n] = S.dh[2 * kDec + n] + y;   // dL/dh3' = carried + residual path
    });
    __syncthreads();
    // 6. GRU layers, top down
    for (int l = 2; l >= 0; --l) {
      if (tid < kDec) {
        const int n = tid;
        const float u = st[kStU + l * kDec + n], c = st[kStC + l * kDec + n];
        const float hp = stp ? stp[kStH + l * kDec + n] : 0.f;
        const float dht = S.dht[n];
        const float du = dht * (hp - c);
        const float dc = dht * (1.f - u);
        const float dcp = dc * (1.f - c * c);
        S.dcp[n] = dcp;
        S.dgp[kDec + n] = du * u * (1.f - u);
        gs[kGsC + l * kDec + n] = dcp;
        gs[kGsG + l * 512 + kDec + n] = du * u * (1.f - u);
      }
      __syncthreads();
      // [d inp ; d(r*h)] = dcp . Wc^T     (wT.cw[l] is (256, 512))
      matvec(w.cw[l], 2 * kDec, kDec, 2 * kDec, S.dcp, S.part, [&](int n, float y) {
        if (n < kDec) {
          S.dinp[n] = y;
        } else {
          const int i = n - kDec;
          const float rr = st[kStR + l * kDec + i];
          const float hp = stp ? stp[kStH + l * kDec + i] : 0.f;
          const float drp = y * hp * rr * (1.f - rr);
          S.dgp[i] = drp;
          gs[kGsG + l * 512 + i] = drp;
          // partial new carry: dht*u + d(rh)*r
          S.dh[l * kDec + i] = S.dht[i] * st[kStU + l * kDec + i] + y * rr;
        }
      });
      __syncthreads();
      // [d inp ; d h] += dgp . Wg^T       (wT.gw[l] is (512, 512))
      matvec(w.gw[l], 2 * kDec, 2 * kDec, 2 * kDec, S.dgp, S.part, [&](int n, float y) {
        if (n < kDec) {
          const float di = S.dinp[n] + y;
          if (l > 0) S.dht[n] = S.dh[(l - 1) * kDec + n] + di;   // next layer down: carried + input path
          else S.dx[n] = S.dy[n] + di;                            // x feeds GRU1 and the residual
        } else {
          S.dh[l * kDec + n - kDec] += y;
        }
      });
      __syncthreads();
    }
    if (tid < kDec) gs[kGsX + tid] = S.dx[tid];
    // 7. input projection: d[p2 ; att_{t-1}] = dx . Wi^T   (wT.in_w is (256, 384))
    matvec(w.in_w, kPre2 + kAtt, kDec, kPre2 + kAtt, S.dx, S.part, [&](int n, float y) {
      if (n < kPre2) {
        const float p2 = st[kStP2 + n];
        const float g = p2 > 0.f ? (a.keep2 ? 2.f * y : y) : 0.f;
        S.dp2[n] = g;
        gs[kGsP2 + n] = g;
      } else {
        S.datt[n - kPre2] = y;
      }
    });
    __syncthreads();
    // 8. pre-net layer 2: dp1 = dp2pre . W2^T   (wT.pre_w2 is (128, 256))
    matvec(w.pre_w2, kPre1, kPre2, kPre1, S.dp2, S.part, [&](int n, float y) {
      const float p1 = st[kStP1 + n];
      const float g = p1 > 0.f ? (a.keep1 ? 2.f * y : y) : 0.f;
      S.dp1[n] = g;
      gs[kGsP1 + n] = g;
    });
    __syncthreads();
    // 9. pre-net layer 1 input gradient, only when this step's input was the previous cell_output
    if (this_from_out) {
      matvec(w.pre_w1, kMel, kPre1, kMel, S.dp1, S.part, [&](int n, float y) { S.dfr[n] = y; });
      __syncthreads();
    }
  }
  // attention_v gradient: reduce per-lane partials across waves, then one atomic per element
  reinterpret_cast<float4*>(S.red + wave * 256)[lane] = dv4;
  __syncthreads();
  if (tid < kAtt) {
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) d += S.red[i * 256 + tid];
    atomicAdd(&a.datt_v[tid], d);
  }
}

}  // namespace

int launch_decoder_fwd(const DecFwdArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.B > 0 && a.Tt > 0 && a.Td > 0 && a.r >= 1 && a.r <= 5, "decoder_fwd: bad dims B=%d Tt=%d Td=%d r=%d", a.B,
               a.Tt, a.Td, a.r);
  const int TtP = (a.Tt + 3) & ~3;
  const size_t smem = (size_t)(kFwdSmemFixed + 2 * TtP) * sizeof(float);
  TACO_REQUIRE(smem <= 160 * 1024, "decoder_fwd: Tt=%d needs %zu bytes of LDS (> 160 KiB)", a.Tt, smem);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      taco_set_error("decoder_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(decoder_fwd_kernel, dim3(a.B), dim3(NT), smem, s, a);
  TACO_LAUNCH_CHECK("decoder_fwd");
  return TACO_OK;
}

int launch_decoder_bwd(const DecBwdArgs& a, hipStream_t s) {
  TACO_REQUIRE(a.B > 0 && a.Tt > 0 && a.Td > 0 && a.r >= 1 && a.r <= 5, "decoder_bwd: bad dims");
  const int TtP = (a.Tt + 3) & ~3;
  const size_t smem = (size_t)(kBwdSmemFixed + 2 * TtP) * sizeof(float);
  TACO_REQUIRE(smem <= 160 * 1024, "decoder_bwd: Tt=%d needs %zu bytes of LDS (> 160 KiB)", a.Tt, smem);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      taco_set_error("decoder_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  hipLaunchKernelGGL(decoder_bwd_kernel, dim3(a.B), dim3(NT), smem, s, a);
  TACO_LAUNCH_CHECK("decoder_bwd");
  return TACO_OK;
}
