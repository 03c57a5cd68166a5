// model.hip -- host-side orchestration of the Tacotron hot path on one GPU + the C ABI (include/taco_hip.h).
// Every function only ENQUEUES work on the caller's stream: no allocation, no synchronisation.
//
//   taco_forward  = Tacotron.inference(train=True) + add_loss_op       (tacotron.py:107-165)
//   taco_backward = opt.compute_gradients(loss)                        (tacotron.py:172)
//   taco_infer    = Tacotron.inference(train=False)                    (tacotron.py:107-154, ops.py:5-25)
#include <cstdarg>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "layout.h"

namespace {

struct Layouts {
  ParamLayout P;
  TransLayout T;
  WsLayout Wtrain, Winfer;
};

const Layouts& layouts_for(const TacoShape& s) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int, int, int>, Layouts*> cache;
  std::lock_guard<std::mutex> g(mu);
  auto key = std::make_tuple(s.B, s.Tt, s.Td, s.r, s.V, s.S > 1 ? s.S : 1);
  auto it = cache.find(key);
  if (it != cache.end()) return *it->second;
  Layouts* L = new Layouts();
  build_param_layout(s, L->P);
  build_trans_layout(s, L->P, L->T);
  build_ws_layout(s, true, L->T, L->Wtrain);
  build_ws_layout(s, false, L->T, L->Winfer);
  cache[key] = L;
  return *L;
}


// ---- HIP-event profiling rings (taco_profile_enable / taco_profile_read[2]) ----
// category: 0 decoder forward kernel, 1 decoder backward kernel, 2 MFMA GEMM family (conv_gemm, gemm_tn, highway stack),
// 3 bi-GRU recurrences.  Every bracketed launch gets a hipEventRecord pair on ITS launch stream plus its algorithmic FLOPs.
struct ProfRing {
  static constexpr int kCap = 4096;
  hipEvent_t start[kCap], stop[kCap];
  double flops[kCap];
  bool armed[kCap];       // the pair rides on the bracketed launch (tail events, common.h) instead of being recorded as two markers
  char label[kCap][96];   // what the launch was (taco_prof_label; read by taco_debug_profile_labels before taco_profile_read2)
  int created = 0;   // events created so far (lazily, in steps: creating 2 x 4096 events up front costs milliseconds)
  int n = 0;
};
ProfRing g_prof[4];
int g_prof_mask = 0;   // bit c: category c is recorded

}  // namespace

int taco_prof_begin(int which, hipStream_t s) {
  if (!(g_prof_mask & (1 << which))) return -1;
  ProfRing& r = g_prof[which];
  if (r.n >= ProfRing::kCap) return -1;
  while (r.created <= r.n) {
    if (hipEventCreate(&r.start[r.created]) != hipSuccess || hipEventCreate(&r.stop[r.created]) != hipSuccess) return -1;
    ++r.created;
  }
  // inside a tail-event scope (common.h) the pair rides on the bracketed launch itself -- two marker packets around the decoder
  // kernels were ~20 us of the timed step; outside (op-level calls, capture) the bracket is two recorded markers as before
  r.armed[r.n] = taco_tail_arm_timing(s, r.start[r.n], r.stop[r.n]);
  if (!r.armed[r.n]) (void)hipEventRecord(r.start[r.n], s);
  r.label[r.n][0] = 0;
  return r.n;
}
void taco_prof_label(int which, int slot, const char* fmt, ...) {
  if (slot < 0) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_prof[which].label[slot], sizeof(g_prof[which].label[slot]), fmt, ap);
  va_end(ap);
}
void taco_prof_end(int which, int slot, hipStream_t s, double flops) {
  if (slot < 0) return;
  ProfRing& r = g_prof[which];
  if (!r.armed[slot]) {
    (void)hipEventRecord(r.stop[slot], s);
  } else {
    // the pair was armed for the next launch on s.  One launch since: it carried both events.  Several (a k-split pair, the decoder
    // in chunks of 32 rows): the first one carried both -- `stop` is recorded again behind the last, which is what counts.  None:
    // both are recorded now.
    const int rode = taco_tail_disarm_timing(s);
    if (rode == 0) (void)hipEventRecord(r.start[slot], s);
    if (rode != 1) (void)hipEventRecord(r.stop[slot], s);
  }
  r.flops[slot] = flops;
  r.n = slot + 1;
}

namespace {
inline int prof_begin(int which, hipStream_t s) { return taco_prof_begin(which, s); }
inline void prof_end(int which, int slot, hipStream_t s) { taco_prof_end(which, slot, s, 0.0); }

// multi-speaker encoder: the fused form (adapters inside the highway stack kernels, round 6) unless TACO_SPK_UNFUSED=1
static bool spk_fused_form(const CbhgP& c) {
  return c.spk && c.has_adapt[0] && c.has_adapt[1] && c.has_adapt[2] && c.has_adapt[3] && !getenv("TACO_SPK_UNFUSED");
}

struct CbhgBufs {
  float *bank, *pool, *pj1pre, *pj1, *pj2pre, *res, *h[5], *hx[4], *th[4], *xg, *out, *ruc, *tapsplit = nullptr;
  int64_t tapsplit_floats = 0;
  float *sv[4], *rowb[4], *h0, *dh0, *dsmall, *dsmall2;   // speaker sites (null without speakers)
  float *dsm[4], *dsm2[5], *dspk_part;                    // their batched backward (round 6)
  const float* spk_e;   // (B,16) gathered speaker embeddings
  float* dspk_e;        // (B,16) their gradient (accumulated)
};
CbhgBufs cbhg_bufs(float* ws, const CbhgWs& w) {
  CbhgBufs b;
  b.bank = ws + w.bank; b.pool = ws + w.pool; b.pj1pre = ws + w.pj1pre; b.pj1 = ws + w.pj1; b.pj2pre = ws + w.pj2pre;
  b.res = ws + w.res;
  for (int l = 0; l < 5; ++l) b.h[l] = ws + w.h[l];
  for (int l = 0; l < 4; ++l) {
    b.hx[l] = ws + w.hx[l];
    b.sv[l] = w.sv[l] >= 0 ? ws + w.sv[l] : nullptr;
    b.rowb[l] = w.rowb[l] >= 0 ? ws + w.rowb[l] : nullptr;
  }
  b.h0 = w.h0 >= 0 ? ws + w.h0 : nullptr;
  b.dh0 = w.dh0 >= 0 ? ws + w.dh0 : nullptr;
  b.dsmall = w.dsmall >= 0 ? ws + w.dsmall : nullptr;
  b.dsmall2 = w.dsmall2 >= 0 ? ws + w.dsmall2 : nullptr;
  for (int l = 0; l < 4; ++l) b.dsm[l] = w.dsm[l] >= 0 ? ws + w.dsm[l] : nullptr;
  for (int l = 0; l < 5; ++l) b.dsm2[l] = w.dsm2[l] >= 0 ? ws + w.dsm2[l] : nullptr;
  b.dspk_part = w.dspk_part >= 0 ? ws + w.dspk_part : nullptr;
  b.spk_e = nullptr;
  b.dspk_e = nullptr;
  for (int l = 0; l < 4; ++l) b.th[l] = ws + w.th[l];
  b.xg = ws + w.xg; b.out = ws + w.out; b.ruc = ws + w.ruc;
  return b;
}

ConvGemmProblem dense_problem(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                              int N, int K, int act) {
  ConvGemmProblem p;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.taps = 1; p.T = M; p.pad_l = 0; p.act = act;
  return p;
}

BiGruWeights bigru_weights(const float* P, const CbhgP& c) {
  BiGruWeights w;
  w.wg[0] = P + c.fw.wg; w.bg[0] = P + c.fw.bg; w.wc[0] = P + c.fw.wc; w.bc[0] = P + c.fw.bc;
  w.wg[1] = P + c.bw.wg; w.bg[1] = P + c.bw.bg; w.wc[1] = P + c.bw.wc; w.bc[1] = P + c.bw.bc;
  return w;
}

// ops.CBHG forward (ops.py:48-132).  x (B*T, cin).
// forward_impl's hook: work to enqueue (on the side stream) at the moment the NEXT cbhg_fwd launches its bi-GRU recurrence -- a
// 64-workgroup kernel of 150-270 us that leaves 192 CUs idle.  Round 6: everything the backward pass derives from the parameters
// alone (transposed weights, their plane images, transposed decoder composites) runs there, beside the post-net recurrence,
// instead of beside the encoder's conv bank and proj1, whose launches it used to slow down by ~30 us.
thread_local std::function<int(hipStream_t)>* g_pre_bigru_hook = nullptr;

int cbhg_fwd(const float* P, const CbhgP& c, const float* x, int B, int T, const CbhgBufs& w, bool keep_ruc,
             hipStream_t s) {
  const int M = B * T, KC = c.K * kCb;
  // conv bank: K 'same' convs + ReLU, concatenated on channels (ops.py:54-62), BN-affine + max-pool(2,1,same) (ops.py:64-71):
  // one batched launch whose epilogue pools along the sequence (gemm2.hip); the un-pooled activations are kept only when a
  // backward pass will read them.  Shapes the DMA kernel does not take run conv and pool as two passes.
  const float bn_rs = 1.0f / sqrtf(1.0f + kBnEps);   // BN in inference mode: gamma / sqrt(moving_var(=1) + eps), folded in the epilogue
  {
    const char* nf = getenv("TACO_NO_POOL_FUSE");   // A/B and test switch: always the two-pass form
    const bool no_fuse = nf && atoi(nf) != 0;
    ConvGemmBatch batch;
    auto fill = [&](bool fused) {
      batch.n = c.K;
      for (int k = 1; k <= c.K; ++k) {
        ConvGemmProblem& p = batch.p[c.K - k];   // widest kernel first: its blocks are the longest-running
        p = ConvGemmProblem();
        p.A = x; p.lda = c.cin; p.W = P + c.bank_w[k - 1]; p.ldw = kCb; p.bias = P + c.bank_b[k - 1];
        p.ldc = KC; p.M = M; p.N = kCb; p.K = c.cin; p.taps = k; p.T = T;
        p.pad_l = (k - 1) / 2; p.act = TACO_ACT_RELU;
        if (fused) {
          p.C = w.pool + (k - 1) * kCb; p.Cpre = keep_ruc ? w.bank + (k - 1) * kCb : nullptr;
          p.scale = P + c.bank_g + (k - 1) * kCb; p.shift = P + c.bank_be + (k - 1) * kCb; p.scale_mul = bn_rs; p.pool = 1;
        } else {
          p.C = w.bank + (k - 1) * kCb;
        }
      }
    };
    int rc = TACO_ENOTFOUND;
    if (!no_fuse) {
      fill(true);
      rc = launch_conv_gemm_batch(batch, s);
    }
    if (rc == TACO_ENOTFOUND) {
      fill(false);
      TACO_TRY(launch_conv_gemm_batch(batch, s));
      TACO_TRY(launch_bn_maxpool(w.bank, P + c.bank_g, P + c.bank_be, w.pool, B, T, KC, s));
    } else {
      TACO_TRY(rc);
    }
  }
  // conv projections (ops.py:75-87) + residual (ops.py:92)
  {
    ConvGemmProblem p;
    p.A = w.pool; p.lda = KC; p.W = P + c.p1_w; p.ldw = c.c1; p.bias = P + c.p1_b; p.scale = P + c.p1_g; p.scale_mul = bn_rs; p.shift = P + c.p1_be;
    p.C = w.pj1; p.Cpre = w.pj1pre; p.ldc = c.c1; p.M = M; p.N = c.c1; p.K = KC; p.taps = 3; p.T = T; p.pad_l = 1;
    p.act = TACO_ACT_RELU;
    TACO_TRY(launch_conv_gemm_tapsplit(p, w.tapsplit, w.tapsplit_floats, s));
  }
  {
    ConvGemmProblem p;
    p.A = w.pj1; p.lda = c.c1; p.W = P + c.p2_w; p.ldw = c.c2; p.bias = P + c.p2_b; p.scale = P + c.p2_g; p.scale_mul = bn_rs; p.shift = P + c.p2_be;
    p.residual = x; p.ldr = c.cin; p.C = w.res; p.Cpre = w.pj2pre; p.ldc = c.c2; p.M = M; p.N = c.c2; p.K = c.c1;
    p.taps = 3; p.T = T; p.pad_l = 1; p.act = TACO_ACT_NONE;
    TACO_TRY(launch_conv_gemm(p, s));
  }
  // highway x4 (ops.py:27-46, 97-107) with the optional input adapter and the per-layer speaker site (ops.py:101-105):
  // concat([h, tile(s)]) . Wa + ba  ==  h . Wa[:128] + (s . Wa[128:] + ba), i.e. a per-sequence bias -- no (B,T,256) concat.
  if (!c.spk) {
    // single-speaker: at most layer 0 has an input adapter (post-net 80 -> 128); the four layers then run as ONE launch
    if (c.has_adapt[0])
      TACO_TRY(launch_conv_gemm(dense_problem(w.h[0], c.c2, P + c.adapt[0].w, kCb, P + c.adapt[0].b, w.hx[0], kCb, M, kCb, c.c2,
                                              TACO_ACT_NONE), s));
    HighwayStackArgs ha;
    ha.x = w.hx[0]; ha.M = M; ha.nl = 4;
    for (int l = 0; l < 4; ++l) {
      ha.wt[l] = P + c.hwT[l].w; ha.bt[l] = P + c.hwT[l].b; ha.wh[l] = P + c.hwH[l].w; ha.bh[l] = P + c.hwH[l].b;
      ha.th[l] = keep_ruc ? w.th[l] : nullptr; ha.y[l] = w.h[l + 1];
    }
    TACO_TRY(launch_highway_stack_fwd(ha, s));
  }
  // TACO_SPK_UNFUSED=1: the per-layer launches of rounds 1-5 (A/B runs, parity of the fused form against them)
  const bool spk_fused = spk_fused_form(c);
  if (spk_fused) {
    // multi-speaker encoder (round 6): the speaker sites of all four layers depend on the speaker embedding alone -- two grouped
    // launches up front (sv_l = relu(dense(spk)), h0 likewise; rowb_l = sv_l . Wa_l[128:] + ba_l) -- and the four layers, adapters
    // included, are ONE launch (highway.hip, adapter form).  Rounds 1-5: ~5 launch-latency-sized launches per layer.
    ConvGemmBatch b1;
    b1.n = 5;
    for (int l = 0; l < 4; ++l)
      b1.p[l] = dense_problem(w.spk_e, 16, P + c.spkd[l].w, kCb, P + c.spkd[l].b, w.sv[l], kCb, B, kCb, 16, TACO_ACT_RELU);
    b1.p[4] = dense_problem(w.spk_e, 16, P + c.gru_init.w, kCb, P + c.gru_init.b, w.h0, kCb, B, kCb, 16, TACO_ACT_RELU);
    TACO_TRY(launch_conv_gemm_batch(b1, s));
    ConvGemmBatch b2;
    b2.n = 4;
    for (int l = 0; l < 4; ++l)
      b2.p[l] = dense_problem(w.sv[l], kCb, P + c.adapt[l].w + (int64_t)kCb * kCb, kCb, P + c.adapt[l].b, w.rowb[l], kCb, B, kCb, kCb,
                              TACO_ACT_NONE);
    TACO_TRY(launch_conv_gemm_batch(b2, s));
    HighwayStackArgs ha;
    ha.x = w.h[0]; ha.M = M; ha.nl = 4; ha.T = T;
    for (int l = 0; l < 4; ++l) {
      ha.wa[l] = P + c.adapt[l].w; ha.rowb[l] = w.rowb[l]; ha.hx[l] = w.hx[l];
      ha.wt[l] = P + c.hwT[l].w; ha.bt[l] = P + c.hwT[l].b; ha.wh[l] = P + c.hwH[l].w; ha.bh[l] = P + c.hwH[l].b;
      ha.th[l] = keep_ruc ? w.th[l] : nullptr; ha.y[l] = w.h[l + 1];
    }
    TACO_TRY(launch_highway_stack_fwd(ha, s));
  }
  for (int l = 0; l < 4 && c.spk && !spk_fused; ++l) {
    if (c.has_adapt[l]) {
      if (c.spk) {
        TACO_TRY(launch_conv_gemm(dense_problem(w.spk_e, 16, P + c.spkd[l].w, kCb, P + c.spkd[l].b, w.sv[l], kCb, B, kCb, 16,
                                                TACO_ACT_RELU), s));
        TACO_TRY(launch_conv_gemm(dense_problem(w.sv[l], kCb, P + c.adapt[l].w + (int64_t)kCb * kCb, kCb, P + c.adapt[l].b,
                                                w.rowb[l], kCb, B, kCb, kCb, TACO_ACT_NONE), s));
        ConvGemmProblem p = dense_problem(w.h[l], kCb, P + c.adapt[l].w, kCb, w.rowb[l], w.hx[l], kCb, M, kCb, kCb, TACO_ACT_NONE);
        p.T = T;
        p.bias_stride = kCb;
        TACO_TRY(launch_conv_gemm(p, s));
      } else {
        TACO_TRY(launch_conv_gemm(dense_problem(w.h[l], c.c2, P + c.adapt[l].w, kCb, P + c.adapt[l].b, w.hx[l], kCb, M, kCb,
                                                c.c2, TACO_ACT_NONE), s));
      }
    }
    ConvGemmBatch batch;
    batch.n = 2;
    batch.p[0] = dense_problem(w.hx[l], kCb, P + c.hwT[l].w, kCb, P + c.hwT[l].b, w.th[l], 2 * kCb, M, kCb, kCb, TACO_ACT_SIGMOID);
    batch.p[1] = dense_problem(w.hx[l], kCb, P + c.hwH[l].w, kCb, P + c.hwH[l].b, w.th[l] + kCb, 2 * kCb, M, kCb, kCb, TACO_ACT_RELU);
    TACO_TRY(launch_conv_gemm_batch(batch, s));
    TACO_TRY(launch_highway_combine(w.th[l], w.hx[l], w.h[l + 1], M, s));
  }
  if (c.spk && !spk_fused)   // bi-GRU initial state of both directions (ops.py:111-124)
    TACO_TRY(launch_conv_gemm(dense_problem(w.spk_e, 16, P + c.gru_init.w, kCb, P + c.gru_init.b, w.h0, kCb, B, kCb, 16,
                                            TACO_ACT_RELU), s));
  // bi-GRU: hoisted x-side projections (4 problems) + persistent recurrence (ops.py:117-128)
  {
    ConvGemmBatch batch;
    batch.n = 4;
    const GruP* g[2] = {&c.fw, &c.bw};
    for (int d = 0; d < 2; ++d) {
      batch.p[2 * d] = dense_problem(w.h[4], kCb, P + g[d]->wg, 2 * kCb, P + g[d]->bg, w.xg + d * 3 * kCb, 6 * kCb, M,
                                     2 * kCb, kCb, TACO_ACT_NONE);
      batch.p[2 * d + 1] = dense_problem(w.h[4], kCb, P + g[d]->wc, kCb, P + g[d]->bc, w.xg + d * 3 * kCb + 2 * kCb,
                                         6 * kCb, M, kCb, kCb, TACO_ACT_NONE);
    }
    TACO_TRY(launch_conv_gemm_batch(batch, s));
  }
  if (g_pre_bigru_hook) {
    std::function<int(hipStream_t)>* h = g_pre_bigru_hook;
    g_pre_bigru_hook = nullptr;
    TACO_TRY((*h)(s));
  }
  TACO_TRY(launch_bigru_fwd(w.xg, bigru_weights(P, c), c.spk ? w.h0 : nullptr, w.out, keep_ruc ? w.ruc : nullptr, B, T, s));
  return TACO_OK;
}

DecWeights dec_weights(const float* P, const ParamLayout& L) {
  DecWeights w;
  w.pre_w1 = P + L.dec_pre1.w; w.pre_b1 = P + L.dec_pre1.b; w.pre_w2 = P + L.dec_pre2.w; w.pre_b2 = P + L.dec_pre2.b;
  w.in_w = P + L.in_proj.w; w.in_b = P + L.in_proj.b;
  for (int l = 0; l < 3; ++l) {
    w.gw[l] = P + L.gru[l].wg; w.gb[l] = P + L.gru[l].bg; w.cw[l] = P + L.gru[l].wc; w.cb[l] = P + L.gru[l].bc;
  }
  w.out_w = P + L.out_proj.w; w.out_b = P + L.out_proj.b;
  w.q_w = P + L.q_w; w.att_v = P + L.att_v; w.att_w = P + L.att_w;
  return w;
}

// Side stream: work that is independent of the main chain runs here while a 64-workgroup recurrent kernel (bi-GRU) or the
// encoder leaves most of the chip idle.  side_fork(): the side stream waits for everything enqueued on `s` so far;
// side_join(): `s` waits for the side work.  No host synchronisation; TACO_NO_OVERLAP=1 keeps everything on `s`.
constexpr int kGradSegments = 5;
struct SideStream {
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_img = nullptr;   // the forward weight images are built (recorded on the side stream; the main stream waits in front of the encoder CBHG)
  bool off = false;
  // gradient-segment events of the most recent taco_backward issued by this thread on this device (taco_wait_grad_segment):
  // segment [4] post-net, [3] decoder, [2] encoder projections / highways / bi-GRU, [1] encoder conv bank, [0] embedding + encoder
  // pre_net of the flat gradient buffer is final
  hipEvent_t ev_seg[kGradSegments] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_seg_use[kGradSegments] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // what taco_wait_grad_segment waits for: ev_seg or a tail event (common.h)
  bool seg_recorded = false;
};
SideStream& side_stream() {
  static thread_local SideStream ss[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  SideStream& x = ss[dev & 15];
  if (!x.side && !x.off) {
    const char* e = getenv("TACO_NO_OVERLAP");
    if ((e && atoi(e) != 0) || hipStreamCreateWithFlags(&x.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&x.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x.ev_join, hipEventDisableTiming) != hipSuccess) {
      x.side = nullptr;
      x.off = true;
    }
  }
  return x;
}
int record_segment(int seg, hipStream_t on) {
  SideStream& x = side_stream();
  if (!x.ev_seg[seg] && hipEventCreateWithFlags(&x.ev_seg[seg], hipEventDisableTiming) != hipSuccess) {
    taco_set_error("taco_backward: cannot create the gradient-segment event");
    return TACO_ELAUNCH;
  }
  // (tail events, common.h: the segment's event is the one riding on the last launch on `on` -- taken out of that stream's ring,
  //  this segment's previous event refills the slot -- instead of a marker behind it)
  bool owned = false;
  if (hipEvent_t t = taco_tail_steal(on, x.ev_seg[seg], &owned)) {
    if (owned) x.ev_seg[seg] = t;
    x.ev_seg_use[seg] = t;   // (not owned: the stop event of a profiling bracket -- taco_wait_grad_segment waits for it before it is bound again)
  } else {
    // no tail event to take (`on` waits for side-stream work behind its last launch: the end of the pass): the marker goes to the
    // SIDE stream, made to wait for `on`'s last launch first -- it covers both streams and sits between no two kernels of `on`
    hipStream_t at = on;
    if (x.side && x.side != on && !x.off && taco_tail_wait(x.side, on)) at = x.side;
    if (hipEventRecord(x.ev_seg[seg], at) != hipSuccess) {
      taco_set_error("taco_backward: hipEventRecord(segment %d) failed", seg);
      return TACO_ELAUNCH;
    }
    x.ev_seg_use[seg] = x.ev_seg[seg];
  }
  if (seg == 0) x.seg_recorded = true;
  return TACO_OK;
}
hipStream_t side_fork(hipStream_t s) {
  SideStream& x = side_stream();
  // (profile bit 4: everything on the caller's stream, so that the per-launch event timing of the GEMM family measures each
  //  kernel by itself instead of two streams' kernels sharing the chip)
  if (x.off || (g_prof_mask & 16)) return s;
  if (taco_tail_wait(x.side, s)) return x.side;
  if (hipEventRecord(x.ev_fork, s) != hipSuccess || hipStreamWaitEvent(x.side, x.ev_fork, 0) != hipSuccess) return s;
  taco_tail_touch(x.side);
  return x.side;
}
int side_join(hipStream_t s, hipStream_t side) {
  if (side == s) return TACO_OK;
  SideStream& x = side_stream();
  if (taco_tail_wait(s, side)) return TACO_OK;
  if (hipEventRecord(x.ev_join, side) != hipSuccess || hipStreamWaitEvent(s, x.ev_join, 0) != hipSuccess) {
    taco_set_error("side_join: event record/wait failed");
    return TACO_ELAUNCH;
  }
  taco_tail_touch(s);
  return TACO_OK;
}

// Decoder composite weights.  The decoder step is a chain of linear maps with few nonlinearities in between; wherever two
// linear maps follow each other (attention layer -> input projection -> GRU-1 gates; output projection -> query layer /
// next step's pre_net) their product is formed here once per call, so that the persistent kernel needs one exchange
// round for the pair instead of two (decoder.hip).  ~0.4 GFLOP per call, two batched launches.
int build_dec_composites(const float* P, const ParamLayout& PL, const WsLayout& W, float* ws, int r, InitBatch& ib, hipStream_t s) {
  const int R80 = kMel * r, KX = kPre2 + R80 + kAtt, NO = dec_out_cols(r);
  const float* Wi = P + PL.in_proj.w;    // (128 + 256, 256)
  const float* Wa = P + PL.att_w;        // (80r + 256, 256)
  const float* Wo = P + PL.out_proj.w;   // (256, 80r)
  const float* Wq = P + PL.q_w;          // (80r, 256)
  const float* Wg0 = P + PL.gru[0].wg;   // (256 + 256, 512)
  float *wx = ws + W.dc_wx, *wg0 = ws + W.dc_wg0, *bg0 = ws + W.dc_bg0, *wo = ws + W.dc_wo, *bo = ws + W.dc_bo,
        *wp1o = ws + W.dc_wp1o, *bp1o = ws + W.dc_bp1o;
  auto chk = [](hipError_t e) {
    if (e != hipSuccess) {
      taco_set_error("build_dec_composites: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
    return TACO_OK;
  };
  const int KA = kPre2 + R80;   // wg0 rows: [0,KA) p2 and out, [KA,KA+256) h1 (= Wg0_h), [KA+256,KA+512) ctx (only used to form VWg)
  (void)chk;
  // copies / zero pads of the composites: jobs of the caller's batched init launch (one kernel instead of six runtime launches)
  TACO_TRY(ib.copy(wx, Wi, (int64_t)kPre2 * kDec));
  TACO_TRY(ib.copy(wg0 + (int64_t)KA * 2 * kDec, Wg0 + (int64_t)kDec * 2 * kDec, (int64_t)kDec * 2 * kDec));
  TACO_TRY(ib.fill2d(wo + kAtt + R80, kDec, NO - kAtt - R80, NO));            // pad columns of [Wo Wq | Wo | 0]
  TACO_TRY(ib.fill(bo + kAtt + R80, NO - kAtt - R80));
  TACO_TRY(ib.copy2d(wo + kAtt, NO, Wo, R80, kDec, R80));
  TACO_TRY(ib.copy(bo + kAtt, P + PL.out_proj.b, R80));
  TACO_TRY(launch_init_batch(ib, s));
  {
    ConvGemmBatch b1;
    b1.n = 5;
    b1.p[0] = dense_problem(Wa, kAtt, Wi + (int64_t)kPre2 * kDec, kDec, nullptr, wx + (int64_t)kPre2 * kDec, kDec, R80 + kAtt, kDec,
                            kAtt, TACO_ACT_NONE);                                                            // Wa Wi_a
    b1.p[1] = dense_problem(Wo, R80, Wq, kAtt, nullptr, wo, NO, kDec, kAtt, R80, TACO_ACT_NONE);             // Wo Wq
    b1.p[2] = dense_problem(P + PL.out_proj.b, R80, Wq, kAtt, nullptr, bo, NO, 1, kAtt, R80, TACO_ACT_NONE); // bo Wq
    b1.p[3] = dense_problem(Wo + (R80 - kMel), R80, P + PL.dec_pre1.w, kPre1, nullptr, wp1o, kPre1, kDec, kPre1, kMel,
                            TACO_ACT_NONE);                                                                  // Wo[:, last frame] W1
    b1.p[4] = dense_problem(P + PL.out_proj.b + (R80 - kMel), R80, P + PL.dec_pre1.w, kPre1, P + PL.dec_pre1.b, bp1o, kPre1, 1,
                            kPre1, kMel, TACO_ACT_NONE);
    TACO_TRY(launch_conv_gemm_batch(b1, s));
    ConvGemmBatch b2;
    b2.n = 3;
    b2.p[0] = dense_problem(wx, kDec, Wg0, 2 * kDec, nullptr, wg0, 2 * kDec, KA, 2 * kDec, kDec, TACO_ACT_NONE);   // Wx_po Wg0_x
    b2.p[1] = dense_problem(P + PL.in_proj.b, kDec, Wg0, 2 * kDec, P + PL.gru[0].bg, bg0, 2 * kDec, 1, 2 * kDec, kDec,
                            TACO_ACT_NONE);                                                                        // bi Wg0_x + bg0
    b2.p[2] = dense_problem(wx + (int64_t)KA * kDec, kDec, Wg0, 2 * kDec, nullptr, wg0 + (int64_t)(KA + kDec) * 2 * kDec, 2 * kDec,
                            kAtt, 2 * kDec, kDec, TACO_ACT_NONE);                                                  // Wx_c Wg0_x
    TACO_TRY(launch_conv_gemm_batch(b2, s));
  }
  return TACO_OK;
}
DecComposite dec_composite(const float* ws, const WsLayout& W, int r) {
  DecComposite c;
  c.wx = ws + W.dc_wx; c.wg0 = ws + W.dc_wg0; c.bg0 = ws + W.dc_bg0; c.wo = ws + W.dc_wo; c.bo = ws + W.dc_bo;
  c.wp1o = ws + W.dc_wp1o; c.bp1o = ws + W.dc_bp1o; c.NO = dec_out_cols(r);
  return c;
}

int prepare_transposes(const float* P, const ParamLayout& L, const TransLayout& T, float* PT, int r, hipStream_t s);
int build_dec_composites_bwd(const float* P, const ParamLayout& PL, const WsLayout& W, float* ws, int r, hipStream_t s);

// One line on stderr, once per shape, when a launch takes decoder.hip instead of decoder3.hip: Config.validate accepts shapes
// (r = 1, 3, 4, Tt > 256) that the fast kernels do not cover, and the fallback costs about 2x per decoder step.
void note_decoder_fallback(int B, int Tt, int r) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int>, bool> seen;
  std::lock_guard<std::mutex> g(mu);
  bool& s = seen[std::make_tuple(B, Tt, r)];
  if (s || getenv("TACO_QUIET")) return;
  s = true;
  const int mode = taco_decoder_mode(-1);
  if (mode >= 2)
    fprintf(stderr, "taco: decoder mode %d (TACO_DEC_V3=0 or escalated after an exchange time-out): decoder.hip runs the decoder "
                    "(about 2x slower per step than decoder3.hip)\n", mode);
  else
    fprintf(stderr, "taco: B=%d Tt=%d r=%d is outside decoder3.hip's scope (Tt <= 256, r in {2, 5}, all 256 workgroups "
                    "co-resident): decoder.hip runs the decoder, about 2x slower per step\n", B, Tt, r);
}

// Pre-split weight images (kernels.h): rebuilds this thread's table for THIS call's parameter / workspace pointers, in the order
// of for_each_weight_image (= the order the workspace region was sized in).  phase 0: forward weights, phase 1: backward
// (transposed) weights; `build`: queue the images of that phase for weight_images_build (taco_forward / taco_infer), or only
// register them (taco_backward: they were built by the taco_forward that ran on this workspace).  TACO_GEMM2_BSPLIT=0: no images.
static int register_weight_images(const Layouts& L, const WsLayout& W, const float* P, float* ws, bool train, int phase, bool build) {
  const char* e = getenv("TACO_GEMM2_BSPLIT");
  if ((e && atoi(e) == 0) || W.wimg < 0) return TACO_OK;
  int rc = TACO_OK;
  int64_t off = 0;
  for_each_weight_image(L.P, L.T, train, [&](int ksrc, int64_t koff, int kld, int dsrc, int64_t doff, int dld, int taps, int K, int N, bool bwd) {
    const int64_t mine = off;
    off += weight_image_floats(taps, K, N);
    if (rc == TACO_OK && weight_image_bytes(taps, K, N) != 4 * weight_image_floats(taps, K, N)) {   // (layout.h and gemm2.hip must agree)
      taco_set_error("weight images: layout.h sizes an image of (%d, %d, %d) at %lld floats, gemm2.hip at %lld bytes", taps, K, N,
                     (long long)weight_image_floats(taps, K, N), (long long)weight_image_bytes(taps, K, N));
      rc = TACO_EINVAL;
    }
    if ((bwd ? 1 : 0) != phase || rc != TACO_OK) return;
    auto at = [&](int src, int64_t o) -> const float* {
      return src == 0 ? P + o : (src == 1 ? ws + W.paramsT + o : ws + W.wd_pad + o);
    };
    const float* key = at(ksrc, koff);
    const float* data = at(dsrc, doff);
    rc = weight_image_add(key, kld, taps, K, N, ws + W.wimg + mine, build, data == key && dld == kld ? nullptr : data, dld);
  });
  return rc;
}

// Tail events (common.h) are tracked while one of these is alive: the C-ABI calls that fork / join streams.  Not while the
// caller's stream is being captured into a graph (a stop event on a captured launch is not a graph dependency).
struct TailScope {
  bool open = false;
  TailScope(hipStream_t s, int kind, const TacoShape& sh) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
      (void)hipGetLastError();   // (not this call's failure to report: launches are checked with hipGetLastError)
      return;
    }
    if (st != hipStreamCaptureStatusNone) return;
    uint64_t key = 1469598103934665603ull;   // (FNV-1a over the call's kind and shape)
    const int64_t f[8] = {kind, sh.B, sh.Tt, sh.Td, sh.r, sh.V, sh.S, (int64_t)(uintptr_t)s};
    for (int64_t v : f) key = (key ^ (uint64_t)v) * 1099511628211ull;
    taco_tail_open(key);
    open = true;
  }
  ~TailScope() { if (open) taco_tail_close(); }
};

// encoder + attention memory + decoder + post-net; shared by train and inference forward.
int forward_impl(const TacoShape& sh, const Layouts& L, const WsLayout& W, const float* P, const int32_t* text,
                 const int32_t* text_length, const int32_t* speaker, const float* mel, const uint8_t* ek1, const uint8_t* ek2, const uint8_t* dk1,
                 const uint8_t* dk2, const uint8_t* sample, float* s2s, float* output, float* align, float* ws, bool train,
                 hipStream_t s) {
  const ParamLayout& PL = L.P;
  const int B = sh.B, Tt = sh.Tt, Td = sh.Td, r = sh.r, R80 = kMel * r;
  const int M1 = B * Tt, M2 = B * Td * r;
  // embedding (tacotron.py:111-114) first: the side stream's fork below then waits for THIS launch's stop event (tail events,
  // common.h) instead of a marker recorded on the caller's stream in front of the call's first kernel
  // (same box, alternated four times: 7.02-7.13 vs 7.07-7.16 ms per step, profiles/r06_fwd_fork_order_ab.txt)
  TACO_TRY(launch_embedding(P + PL.emb, text, ws + W.emb, M1, sh.V, s));
  // decoder composites depend on the parameters only: side stream, concurrent with the encoder
  hipStream_t sd = side_fork(s);
  // pre-split bf16 plane images of the forward weights (gemm2.hip's B-image form): first thing on the side stream, beside the
  // embedding gather and the encoder pre_net; the main stream waits for them in front of the encoder CBHG (its first gemm2 launch)
  weight_images_clear();
  TACO_TRY(register_weight_images(L, W, P, ws, train, 0, true));
  TACO_TRY(weight_images_build(sd));
  bool img_event = false;
  hipEvent_t img_ev = nullptr;
  if (sd != s) {
    SideStream& x = side_stream();
    // (tail events: the image launch's own event.  Its ring slot comes round again after 64 more launches on the side stream --
    //  far more than are enqueued before the wait below -- and would then name a LATER launch of the same stream: still correct)
    img_ev = taco_tail_event(sd, s);
    if (img_ev) {
      img_event = true;
    } else {
      if (!x.ev_img && hipEventCreateWithFlags(&x.ev_img, hipEventDisableTiming) != hipSuccess) x.ev_img = nullptr;
      if (x.ev_img && hipEventRecord(x.ev_img, sd) == hipSuccess) img_event = true, img_ev = x.ev_img;
      else TACO_TRY(side_join(s, sd));   // (no event: fall back to a full join here)
    }
  }
  // ONE batched init launch for everything that is a plain copy / zero pad of parameters (side stream, first thing on it):
  //   post/dense (256, 1025) kernel -> pitch 1028 so the W tile loads are 16-byte aligned float4 (pad columns are never stored);
  //   the copied / zero-padded parts of the decoder composites; the pad rows / columns of two transposed backward operands
  InitBatch ib;
  TACO_TRY(ib.copy2d(ws + W.wd_pad, 1028, P + PL.post_dense.w, kFft, 2 * kCb, kFft));
  if (train) {
    TACO_TRY(ib.fill(ws + W.paramsT + L.T.post_dense + (int64_t)kFft * 2 * kCb, 3 * 2 * kCb));
    const int NF = dec_fan_cols(r);
    TACO_TRY(ib.fill2d(ws + W.bc_fa + R80, kDec, NF - R80, NF));
  }
  // the decoder's exchange area (granule epochs of the previous launch): zeroed by the same launch; decoder3.hip then issues no
  // memset in front of the forward decoder (decoder.hip, the fallback, still zeroes for itself)
  TACO_TRY(ib.fill(ws + W.xchg, decoder_xchg_bytes(B, Tt) / 4));
  // pre-net input frames of the TEACHER-FORCED steps (the layer-1 weight gradient reads W.prein for every step): the last frame of
  // mel[t], known now -- copied here instead of being loaded and stored frame by frame by the lead workgroup of every decoder
  // cluster behind round E (the lead is on every peer's critical path).  decoder3.hip writes only the frames of steps that are fed
  // by the previous output (scheduled sampling); decoder.hip, the fallback, still writes all of them itself.
  if (train) TACO_TRY(ib.copy2d(ws + W.prein, kMel, mel + kMel * (r - 1), R80, B * Td, kMel));
  TACO_TRY(build_dec_composites(P, PL, W, ws, r, ib, sd));
  if (train) {
    // everything the backward pass derives from the parameters alone (transposed / tap-flipped weight copies, transposed
    // composites) is built here, beside the encoder, instead of at the head of taco_backward's critical path
    // (round 6: enqueued beside the POST-NET bi-GRU recurrence instead -- `bwd_prep` below; TACO_BWD_PREP_EARLY=1: here, as in rounds 2-5)
    if (getenv("TACO_BWD_PREP_EARLY")) {
      TACO_TRY(prepare_transposes(P, PL, L.T, ws + W.paramsT, r, sd));
      TACO_TRY(register_weight_images(L, W, P, ws, train, 1, true));
      TACO_TRY(weight_images_build(sd));
      TACO_TRY(build_dec_composites_bwd(P, PL, W, ws, r, sd));
    }
    // decoder pre_net (tacotron.py:38-44, 64-71) of every TEACHER-FORCED step: its input (the last frame of mel[t]) is known now,
    // so the two layers are two GEMMs over all B*Td frames, straight into the P1 / P2 slots of the decoder stash, beside the
    // encoder.  The decoder kernel loads P2 on teacher-forced steps and runs (and overwrites) the pre-net only where a step is
    // fed by the previous output (scheduled sampling).
    {
      const int MDf = B * Td;
      ConvGemmProblem q1 = dense_problem(mel + kMel * (r - 1), R80, P + PL.dec_pre1.w, kPre1, P + PL.dec_pre1.b,
                                         ws + W.stash + kStP1, kStRec, MDf, kPre1, kMel, TACO_ACT_RELU);
      q1.keep = dk1;
      TACO_TRY(launch_conv_gemm(q1, sd));
      ConvGemmProblem q2 = dense_problem(ws + W.stash + kStP1, kStRec, P + PL.dec_pre2.w, kPre2, P + PL.dec_pre2.b,
                                         ws + W.stash + kStP2, kStRec, MDf, kPre2, kPre1, TACO_ACT_RELU);
      q2.keep = dk2;
      TACO_TRY(launch_conv_gemm(q2, sd));
    }
  }
  // encoder pre_net (tacotron.py:128) on the embedding gathered above
  if (!getenv("TACO_NO_PRENET_FUSE")) {   // both layers in one launch (prenet.hip)
    PrenetArgs pa;
    pa.x = ws + W.emb; pa.w1 = P + PL.enc_pre1.w; pa.b1 = P + PL.enc_pre1.b; pa.w2 = P + PL.enc_pre2.w; pa.b2 = P + PL.enc_pre2.b;
    pa.keep1 = train ? ek1 : nullptr; pa.keep2 = train ? ek2 : nullptr;
    pa.y1 = ws + W.p1; pa.y2 = ws + W.p2; pa.M = M1;
    pa.trace = getenv("TACO_PN_TRACE") ? reinterpret_cast<long long*>(ws + W.err + 400) : nullptr;
    TACO_TRY(launch_prenet_fwd(pa, s));
  } else {
    ConvGemmProblem p = dense_problem(ws + W.emb, kEmbed, P + PL.enc_pre1.w, kPre1, P + PL.enc_pre1.b, ws + W.p1, kPre1, M1,
                                      kPre1, kEmbed, TACO_ACT_RELU);
    p.keep = train ? ek1 : nullptr;
    TACO_TRY(launch_conv_gemm(p, s));
    ConvGemmProblem q = dense_problem(ws + W.p1, kPre1, P + PL.enc_pre2.w, kPre2, P + PL.enc_pre2.b, ws + W.p2, kPre2, M1,
                                      kPre2, kPre1, TACO_ACT_RELU);
    q.keep = train ? ek2 : nullptr;
    TACO_TRY(launch_conv_gemm(q, s));
  }
  CbhgBufs eb = cbhg_bufs(ws, W.enc);
  eb.tapsplit = ws + W.tapsplit;
  eb.tapsplit_floats = W.tapsplit_floats;
  if (PL.enc.spk) {   // speaker embedding lookup (tacotron.py:117-124)
    TACO_REQUIRE(speaker != nullptr, "num_speakers=%d but no speaker ids were given", sh.S);
    TACO_TRY(launch_embedding(P + PL.spk_embed, speaker, ws + W.spk_e, B, sh.S, s, 16));
    eb.spk_e = ws + W.spk_e;
  }
  if (img_event) {
    if (hipStreamWaitEvent(s, img_ev, 0) != hipSuccess) {
      taco_set_error("forward: cannot wait for the weight images");
      return TACO_ELAUNCH;
    }
    taco_tail_touch(s);
  }
  TACO_TRY(cbhg_fwd(P, PL.enc, ws + W.p2, B, Tt, eb, train, s));
  // attention memory (BahdanauAttention.__init__; tacotron.py:48-52)
  TACO_TRY(launch_mask_rows(eb.out, text_length, ws + W.values, B, Tt, kAtt, s));
  // (keys = values . Wm rides in the same grouped launch as the context folds below: three products of `values`)
  // decoder (tacotron.py:134-138)
  TACO_TRY(side_join(s, sd));
  DecFwdArgs da;
  da.w = dec_weights(P, PL);
  da.c = dec_composite(ws, W, r);
  {
    // context = alignments . values only ever enters the next step through Wx_c (and Wx_c Wg0_x): fold it per memory row, once
    // per call, so that the decoder step needs no context mat-vec / all-gather round (decoder.hip)
    const int KA = kPre2 + R80;
    if (train) {
      // backward twin: the same fold of the CENTRED memory rows (d alignments are only defined up to a per-step constant;
      // centring first keeps the rounding of each row's fold relative to what survives the softmax backward).  Only
      // taco_backward reads it: side stream, joined at the end of the forward pass.
      hipStream_t sv = side_fork(s);
      TACO_TRY(launch_center_rows(ws + W.values, text_length, ws + W.vcen, B, Tt, kAtt, sv));
      TACO_TRY(launch_conv_gemm(dense_problem(ws + W.vcen, kAtt, da.c.wx + (int64_t)KA * kDec, kDec, nullptr, ws + W.vwxc, kDec, M1,
                                              kDec, kAtt, TACO_ACT_NONE), sv));
    }
    ConvGemmBatch vb;
    vb.n = 3;
    vb.p[2] = dense_problem(ws + W.values, kAtt, P + PL.mem_w, kAtt, nullptr, ws + W.keys, kAtt, M1, kAtt, 2 * kCb, TACO_ACT_NONE);
    vb.p[0] = dense_problem(ws + W.values, kAtt, da.c.wx + (int64_t)KA * kDec, kDec, nullptr, ws + W.vwx, kDec, M1, kDec, kAtt,
                            TACO_ACT_NONE);
    vb.p[1] = dense_problem(ws + W.values, kAtt, da.c.wg0 + (int64_t)(KA + kDec) * 2 * kDec, 2 * kDec, nullptr, ws + W.vwg, 2 * kDec,
                            M1, 2 * kDec, kAtt, TACO_ACT_NONE);
    TACO_TRY(launch_conv_gemm_batch(vb, s));
  }
  da.keys = ws + W.keys; da.values = ws + W.values; da.vwx = ws + W.vwx; da.vwg = ws + W.vwg; da.text_length = text_length;
  da.mel = train ? mel : nullptr;
  da.keep1 = train ? dk1 : nullptr; da.keep2 = train ? dk2 : nullptr; da.sample = train ? sample : nullptr;
  da.out = s2s; da.align = align;
  da.stash = train ? ws + W.stash : nullptr;
  da.prein = train ? ws + W.prein : nullptr;
  da.pre2 = train ? ws + W.stash + kStP2 : nullptr;
  da.ldpre2 = kStRec;
  da.xchg = ws + W.xchg; da.err = reinterpret_cast<int*>(ws + W.err);
  da.trace = getenv("TACO_DEC_TRACE") ? reinterpret_cast<long long*>(ws + W.err + 16) : nullptr;
  da.B = B; da.Tt = Tt; da.Td = Td; da.r = r; da.P = 1;
  // (the error words are STICKY: only taco_clear_error() resets them, so a time-out in any step stays visible to the
  //  guarded Adam update and to the host's next check, however rarely the host looks)
  {
    const int slot = prof_begin(0, s);
    da.xchg_zeroed = 1;
    int rc = launch_decoder3_fwd(da, s);
    if (rc == TACO_ENOTFOUND) {
      note_decoder_fallback(B, Tt, r);
      rc = launch_decoder_fwd(da, s);
    }
    TACO_TRY(rc);
    prof_end(0, slot, s);
  }
  hipStream_t sl = s;
  if (train) {
    // seq2seq half of add_loss_op (tacotron.py:158) only needs the decoder output: side stream, beside the post-net
    sl = side_fork(s);
    TACO_TRY(launch_l1(s2s, mel, ws + W.ds2s, R80, ws + W.loss + 4, (int64_t)B * Td, R80, sl));
  }
  // post-net (tacotron.py:142-152): (B,Td,80r) reinterpreted as (B, Td*r, 80)
  CbhgBufs pb = cbhg_bufs(ws, W.post);
  pb.tapsplit = ws + W.tapsplit;
  pb.tapsplit_floats = W.tapsplit_floats;
  // transposed / tap-flipped weight copies, their plane images (B operand of the backward GEMMs; same stream, behind the copies) and
  // the transposed decoder composites: side stream, forked where the post-net's bi-GRU recurrence starts
  std::function<int(hipStream_t)> bwd_prep = [&](hipStream_t at) -> int {
    hipStream_t q = side_fork(at);
    TACO_TRY(prepare_transposes(P, PL, L.T, ws + W.paramsT, r, q));
    TACO_TRY(register_weight_images(L, W, P, ws, train, 1, true));
    TACO_TRY(weight_images_build(q));
    TACO_TRY(build_dec_composites_bwd(P, PL, W, ws, r, q));
    if (q != at && sl == s) sl = q;   // (make sure the join below covers it)
    return TACO_OK;
  };
  const bool late_prep = train && !getenv("TACO_BWD_PREP_EARLY");
  if (late_prep) g_pre_bigru_hook = &bwd_prep;
  const int rc_post = cbhg_fwd(P, PL.post, s2s, B, Td * r, pb, train, s);
  if (g_pre_bigru_hook) {   // (not reached: cbhg_fwd failed before its recurrence)
    g_pre_bigru_hook = nullptr;
  }
  TACO_TRY(rc_post);
  {
    // output rows are 1025 floats apart: gemm2.hip writes them with its shifted float4 epilogue (92 vs 105 us with scalar stores)
    ConvGemmProblem p = dense_problem(pb.out, 2 * kCb, ws + W.wd_pad, 1028, P + PL.post_dense.b, output, kFft, M2, kFft,
                                      2 * kCb, TACO_ACT_NONE);
    p.Nld = 1028;
    TACO_TRY(launch_conv_gemm(p, s));
  }
  TACO_TRY(side_join(s, sl));
  return TACO_OK;
}

// Weight-gradient GEMM.  Inside a TnGroup scope the call is queued and launched with its independent siblings in ONE grid
// (launch_gemm_tn_batch); otherwise it is launched immediately.
thread_local GemmTnBatch* g_tnq = nullptr;
// Side routing: while set, every weight-gradient launch issued for stream `s` goes to this stream instead, ordered behind the
// work enqueued on `s` so far by an event.  The CBHG backward passes use it: their ~27 weight-gradient GEMMs (0.5 / 0.7 ms per
// step) feed nothing but the gradient buffer, so they run beside the activation-gradient chain -- much of which is small
// launches that leave most CUs idle -- instead of inside it.  Their operands then must not be reused in place by the chain
// (BwdScratch::alt_*).  TACO_NO_SIDE_TN=1 keeps them on the main stream (A/B runs).
thread_local hipStream_t g_tn_side = nullptr;
thread_local hipEvent_t g_tn_ev = nullptr;
int tn_route(hipStream_t s, hipStream_t* out) {
  *out = s;
  if (!g_tn_side || g_tn_side == s || (g_prof_mask & 16)) return TACO_OK;
  if (!g_tn_ev && hipEventCreateWithFlags(&g_tn_ev, hipEventDisableTiming) != hipSuccess) {
    taco_set_error("weight-gradient side stream: cannot create an event");
    return TACO_ELAUNCH;
  }
  if (taco_tail_wait(g_tn_side, s)) {
    *out = g_tn_side;
    return TACO_OK;
  }
  if (hipEventRecord(g_tn_ev, s) != hipSuccess || hipStreamWaitEvent(g_tn_side, g_tn_ev, 0) != hipSuccess) {
    taco_set_error("weight-gradient side stream: event record/wait failed");
    return TACO_ELAUNCH;
  }
  taco_tail_touch(g_tn_side);
  *out = g_tn_side;
  return TACO_OK;
}
int tn_launch_batch(GemmTnBatch& b, hipStream_t s) {
  hipStream_t q;
  TACO_TRY(tn_route(s, &q));
  return launch_gemm_tn_batch(b, q);
}
struct TnGroup {
  GemmTnBatch batch;
  hipStream_t s;
  explicit TnGroup(hipStream_t stream) : s(stream) { g_tnq = &batch; }
  int flush() { return batch.n ? tn_launch_batch(batch, s) : TACO_OK; }
  ~TnGroup() { g_tnq = nullptr; }
};

int tn(const float* A, int lda, int K, const float* Y, int ldy, int N, float* W, int ldw, int M, int T, int pad_l,
       hipStream_t s, int taps = 1, float* dbias = nullptr, int Nld = 0) {
  GemmTnArgs a;
  a.dbias = dbias;
  a.Nld = Nld;
  a.A = A; a.lda = lda; a.Y = Y; a.ldy = ldy; a.W = W; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.taps = taps; a.T = T;
  a.pad_l = pad_l;
  if (g_tnq) {
    if (g_tnq->n == kMaxTnBatch) TACO_TRY(tn_launch_batch(*g_tnq, s));
    g_tnq->p[g_tnq->n++] = a;
    return TACO_OK;
  }
  hipStream_t q;
  TACO_TRY(tn_route(s, &q));
  return launch_gemm_tn(a, false, q);
}

// Builds every transposed / flipped weight copy the backward pass needs.
int prepare_transposes(const float* P, const ParamLayout& L, const TransLayout& T, float* PT, int r, hipStream_t s) {
  const int R80 = kMel * r;
  TransposeBatch tb;
  auto tr = [&](int64_t src, int64_t dst, int taps, int K, int N) {
    TACO_REQUIRE(tb.n < kMaxTransposeBatch, "prepare_transposes: too many jobs");
    TransposeJob& j = tb.j[tb.n++];
    j.in = P + src; j.out = PT + dst; j.taps = taps; j.K = K; j.N = N; j.tile0 = 0;
    return TACO_OK;
  };
  TACO_TRY(tr(L.enc_pre1.w, T.enc_pre1, 1, kEmbed, kPre1));
  TACO_TRY(tr(L.enc_pre2.w, T.enc_pre2, 1, kPre1, kPre2));
  const CbhgP* cp[2] = {&L.enc, &L.post};
  const CbhgT* ct[2] = {&T.enc, &T.post};
  for (int i = 0; i < 2; ++i) {
    const CbhgP& c = *cp[i];
    const CbhgT& t = *ct[i];
    for (int k = 1; k <= c.K; ++k) TACO_TRY(tr(c.bank_w[k - 1], t.bank[k - 1], k, c.cin, kCb));
    TACO_TRY(tr(c.p1_w, t.p1, 3, c.K * kCb, c.c1));
    TACO_TRY(tr(c.p2_w, t.p2, 3, c.c1, c.c2));
    for (int l = 0; l < 4; ++l) {
      if (!c.has_adapt[l]) continue;
      if (c.spk) {
        TACO_TRY(tr(c.adapt[l].w, t.adapt[l], 1, kCb, kCb));
        TACO_TRY(tr(c.adapt[l].w + (int64_t)kCb * kCb, t.adapt_s[l], 1, kCb, kCb));
        TACO_TRY(tr(c.spkd[l].w, t.spkd[l], 1, 16, kCb));
      } else {
        TACO_TRY(tr(c.adapt[l].w, t.adapt[l], 1, c.c2, kCb));
      }
    }
    if (c.spk) TACO_TRY(tr(c.gru_init.w, t.gru_init, 1, 16, kCb));
    for (int l = 0; l < 4; ++l) {
      TACO_TRY(tr(c.hwT[l].w, t.hw[l], 1, kCb, kCb));
      TACO_TRY(tr(c.hwH[l].w, t.hw[l] + kCb * kCb, 1, kCb, kCb));
    }
    const GruP* g[2] = {&c.fw, &c.bw};
    for (int d = 0; d < 2; ++d) {
      // x-parts: rows [0,128) of gates (128x256) and candidate (128x128) kernels
      TACO_TRY(tr(g[d]->wg, t.gru_x + (int64_t)d * 3 * kCb * kCb, 1, kCb, 2 * kCb));
      TACO_TRY(tr(g[d]->wc, t.gru_x + (int64_t)d * 3 * kCb * kCb + 2 * kCb * kCb, 1, kCb, kCb));
      // h-parts: rows [128,256)
      TACO_TRY(tr(g[d]->wg + (int64_t)kCb * 2 * kCb, t.wghT[d], 1, kCb, 2 * kCb));
      TACO_TRY(tr(g[d]->wc + (int64_t)kCb * kCb, t.wchT[d], 1, kCb, kCb));
    }
  }
  TACO_TRY(tr(L.mem_w, T.mem_w, 1, 2 * kCb, kAtt));
  TACO_TRY(tr(L.dec_pre1.w, T.dec_pre1, 1, kMel, kPre1));
  TACO_TRY(tr(L.dec_pre2.w, T.dec_pre2, 1, kPre1, kPre2));
  TACO_TRY(tr(L.in_proj.w, T.in_proj, 1, kPre2 + kAtt, kDec));
  for (int l = 0; l < 3; ++l) {
    TACO_TRY(tr(L.gru[l].wg, T.gw[l], 1, 2 * kDec, 2 * kDec));
    TACO_TRY(tr(L.gru[l].wc, T.cw[l], 1, 2 * kDec, kDec));
  }
  TACO_TRY(tr(L.out_proj.w, T.out_proj, 1, kDec, R80));
  TACO_TRY(tr(L.q_w, T.q_w, 1, R80, kAtt));
  TACO_TRY(tr(L.att_w, T.att_w, 1, R80 + kAtt, kAtt));
  TACO_TRY(tr(L.post_dense.w, T.post_dense, 1, 2 * kCb, kFft));
  // (the three zero pad rows behind the transposed post/dense kernel are written by forward_impl's batched init launch)
  return launch_transpose_batch(tb, s);
}

// Transposed composites of the decoder backward kernel (from the forward composites, still in the workspace since
// taco_forward, and the out-projection kernel).
int build_dec_composites_bwd(const float* P, const ParamLayout& PL, const WsLayout& W, float* ws, int r, hipStream_t s) {
  const int R80 = kMel * r, NO = dec_fan_cols(r);
  float *fa = ws + W.bc_fa, *wot = ws + W.bc_wot;
  // (fa's pad columns [80r, NO) are zeroed by forward_impl's batched init launch; columns [0, 80r) are written below)
  TransposeBatch tb;
  auto job = [&](const float* in, int ldi, float* out, int ldo, int K, int N) {
    TransposeJob& j = tb.j[tb.n++];
    j.in = in; j.out = out; j.taps = 1; j.K = K; j.N = N; j.tile0 = 0; j.ldi = ldi; j.ldo = ldo;
  };
  const float* wx = ws + W.dc_wx;
  job(P + PL.mem_w, kAtt, ws + W.bc_wmx, 2 * kCb, 2 * kCb, kAtt);                     // Wm^T     } stacked: [d keys | E] . [Wm^T ; Wx_c^T]
  job(wx + (int64_t)(kPre2 + R80) * kDec, kDec, ws + W.bc_wxct, kAtt, kAtt, kDec);   // Wx_c^T   }
  job(wx + (int64_t)kPre2 * kDec, kDec, fa, NO, R80, kDec);                          // Wx_o^T -> fa[:, 0:80r]
  job(P + PL.out_proj.w, R80, wot, kDec, kDec, R80);                                 // Wo^T
  job(ws + W.dc_wo, dec_out_cols(r), wot + (int64_t)R80 * kDec, kDec, kDec, kAtt);   // (Wo Wq)^T  (dc_wo is [Wo Wq | Wo | 0], pitch dec_out_cols)
  job(ws + W.dc_wp1o, kPre1, wot + (int64_t)(R80 + kAtt) * kDec, kDec, kDec, kPre1); // (Wo_f W1)^T
  TACO_TRY(launch_transpose_batch(tb, s));
  // dx_{t+1} -> d cell_output_t -> d(x + h3)_t without the stop in between: Wx_o^T Wo^T = fa[:, 0:80r] . wot[0:80r, :]
  return launch_conv_gemm(dense_problem(fa, NO, wot, kDec, nullptr, ws + W.bc_wdx, kDec, kDec, kDec, R80, TACO_ACT_NONE), s);
}

struct BwdScratch {
  float *gA, *gB, *gC, *gD, *gE, *gF, *gG;
  // When the weight-gradient GEMMs run on the side stream their operands must outlive the chain's in-place reuse: these three
  // buffers then replace gD / gC / gA (d pj1, d z1, d pool); null = reuse as before.
  float *alt_dpj1 = nullptr, *alt_dz1 = nullptr, *alt_dpool = nullptr;
  bool dx_zeroed = false;   // dx_out was zeroed by the caller (taco_backward's batched init launch)
};

// CBHG backward.  dOut (M,256) -> dX (M,cin) written to `dx_out`; parameter gradients accumulated into G.
// x is the CBHG input.  Scratch: gA/gB (M, K*128), gC (M,768), gD..gG (M,256).
// seg_after_p1 >= 0: that gradient segment (projections, highways, bi-GRU -- everything of this CBHG except the conv bank) is
// announced as soon as its last contributor, proj1's weight gradient, has been enqueued.
// seg_after_bank >= 0 (round 6, late): likewise the conv bank's own range [bank_w[0], p1_w) behind its grouped weight-gradient launch
// (its BN gradients come out of the pooled backward epilogue before that): 8.9 MB of the encoder's last segment no longer wait for the
// step's tail (bank input gradient, pre_net chain, embedding scatter) before they can travel.
int cbhg_bwd(const float* P, const float* PT, float* G, const CbhgP& c, const CbhgT& t, const float* x, const float* dOut,
             int B, int T, const CbhgBufs& w, const BwdScratch& sc, float* dx_out, int seg_after_p1, int seg_after_bank, hipStream_t s) {
  const int M = B * T, KC = c.K * kCb;
  // ---- bi-GRU ----
  float* dxg = sc.gC;   // (M,768)
  float* rh = sc.gD;    // (M,256)
  BiGruBwdWeights bw;
  for (int d = 0; d < 2; ++d) {
    bw.wghT[d] = PT + t.wghT[d];
    bw.wchT[d] = PT + t.wchT[d];
  }
  TACO_TRY(launch_bigru_bwd(dOut, w.out, w.ruc, bw, c.spk ? w.h0 : nullptr, dxg, rh, c.spk ? w.dh0 : nullptr, B, T, s));
  const GruP* g[2] = {&c.fw, &c.bw};
  TnGroup gru_group(s);   // the 8 (10 with speakers) bi-GRU weight gradients are independent: one grid
  for (int d = 0; d < 2; ++d) {
    const float* dG = dxg + d * 3 * kCb;
    const float* dC = dG + 2 * kCb;
    TACO_TRY(tn(w.h[4], kCb, kCb, dG, 6 * kCb, 2 * kCb, G + g[d]->wg, 2 * kCb, M, T, 0, s, 1, G + g[d]->bg));
    TACO_TRY(tn(w.out + d * kCb, 2 * kCb, kCb, dG, 6 * kCb, 2 * kCb, G + g[d]->wg + (int64_t)kCb * 2 * kCb, 2 * kCb, M, T,
                d == 0 ? 1 : -1, s));
    TACO_TRY(tn(w.h[4], kCb, kCb, dC, 6 * kCb, kCb, G + g[d]->wc, kCb, M, T, 0, s, 1, G + g[d]->bc));
    TACO_TRY(tn(rh + d * kCb, 2 * kCb, kCb, dC, 6 * kCb, kCb, G + g[d]->wc + (int64_t)kCb * kCb, kCb, M, T, 0, s));
    if (c.spk) {
      // non-zero initial state: the first step of each direction sees h_prev = h0 (the shifted-row GEMM above treats it as 0)
      const float* dG0 = dG + (int64_t)(d == 0 ? 0 : T - 1) * 6 * kCb;   // row t_first of every sequence, stride T*768
      TACO_TRY(tn(w.h0, kCb, kCb, dG0, T * 6 * kCb, 2 * kCb, G + g[d]->wg + (int64_t)kCb * 2 * kCb, 2 * kCb, B, B, 0, s));
    }
  }
  TACO_TRY(gru_group.flush());
  g_tnq = nullptr;
  // small-tensor helper for the speaker sites: dz (B,128) -> dW (16,128), db, dspk_e (B,16) += dz . W^T
  auto spk_dense_bwd = [&](const float* dz, const DenseP& dp, int64_t wT) -> int {
    TACO_TRY(tn(w.spk_e, 16, 16, dz, kCb, kCb, G + dp.w, kCb, B, B, 0, s, 1, G + dp.b));
    ConvGemmProblem p = dense_problem(dz, kCb, PT + wT, 16, nullptr, w.dspk_e, 16, B, 16, kCb, TACO_ACT_NONE);
    p.residual = w.dspk_e;
    p.ldr = 16;
    return launch_conv_gemm(p, s);
  };
  const bool spk_fused = spk_fused_form(c);
  if (spk_fused) {
    // d h0 = relu'(h0) (dh0[fw] + dh0[bw]) goes into slot 4 of the (5,B,128) gradient block; the ReLU backward of all five slots
    // and everything behind it is batched after the highway chain below
    TACO_TRY(launch_add(w.dh0, w.dh0 + (int64_t)B * kCb, w.dsm2[4], (int64_t)B * kCb, s));
  } else if (c.spk) {
    hipError_t e = hipMemsetAsync(w.dspk_e, 0, (size_t)B * 16 * sizeof(float), s);
    taco_tail_touch(s);
    if (e != hipSuccess) {
      taco_set_error("cbhg_bwd: memset: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
    // h0 = relu(dense(spk)) feeds both directions: d h0 = dh0[fw] + dh0[bw]
    TACO_TRY(launch_add(w.dh0, w.dh0 + (int64_t)B * kCb, w.dsmall, (int64_t)B * kCb, s));
    TACO_TRY(launch_act_bwd(w.h0, w.dsmall, nullptr, w.dsmall, (int64_t)B * kCb, TACO_ACT_RELU, s));
    TACO_TRY(spk_dense_bwd(w.dsmall, c.gru_init, t.gru_init));
  }
  float* gh = sc.gE;     // (M,128) gradient wrt current highway output
  float* gh2 = sc.gF;    // ping-pong
  {
    // d h4 = dxg . Wx^T: (M, 768) x (768, 128) is 50 / 90 tiles of 128 x 128 -- too few for the DMA kernel, and as 64 x 64 tiles of the
    // small kernel one wave of workgroups that each walk all 768 k (37 / 49 us on an idle chip, 55-105 us beside the weight
    // gradients).  Cut along k instead (launch_conv_gemm_tapsplit: S grouped chunks + ordered slab sum; the slab area is free until
    // the bank gather at the end of this pass).  TACO_XPROJ_BWD_KSPLIT=0: one launch, as in rounds 1-6.
    const ConvGemmProblem q = dense_problem(dxg, 6 * kCb, PT + t.gru_x, kCb, nullptr, gh, kCb, M, kCb, 6 * kCb, TACO_ACT_NONE);
    const char* e = getenv("TACO_XPROJ_BWD_KSPLIT");
    if (w.tapsplit && !(e && atoi(e) == 0) && !taco_deterministic()) TACO_TRY(launch_conv_gemm_tapsplit(q, w.tapsplit, w.tapsplit_floats, s));
    else TACO_TRY(launch_conv_gemm(q, s));
  }
  // ---- highway layers 3..0 (with their input adapters / speaker sites) ----
  // d[T|H] of every layer gets its own (M,256) slice of gA (free until dpool below), so the T / H weight gradients of all
  // four layers can wait for ONE grouped launch after the loop; only adapter layers, whose operands live in the ping-pong
  // buffers, flush early.
  float* dxd = sc.gG;    // (M,128)
  if (!c.spk) {
    // single-speaker: the activation-gradient chain of the four layers is ONE launch (highway.hip), their T / H weight
    // gradients one grouped launch; only layer 0 may have an input adapter (post-net 80 -> 128), handled after it
    HighwayStackBwdArgs hb;
    hb.g = gh; hb.gout = gh2; hb.M = M; hb.nl = 4;
    for (int l = 0; l < 4; ++l) {
      hb.wT[l] = PT + t.hw[l]; hb.th[l] = w.th[l]; hb.x[l] = w.hx[l];
      hb.dth[l] = sc.gA + (int64_t)l * M * 2 * kCb;
    }
    TACO_TRY(launch_highway_stack_bwd(hb, s));
    TnGroup hw_group(s);
    for (int l = 0; l < 4; ++l) {
      TACO_TRY(tn(w.hx[l], kCb, kCb, hb.dth[l], 2 * kCb, kCb, G + c.hwT[l].w, kCb, M, M, 0, s, 1, G + c.hwT[l].b));
      TACO_TRY(tn(w.hx[l], kCb, kCb, hb.dth[l] + kCb, 2 * kCb, kCb, G + c.hwH[l].w, kCb, M, M, 0, s, 1, G + c.hwH[l].b));
    }
    if (c.has_adapt[0]) {   // gh2 = d hx[0]: adapter weight / bias gradients and d h[0] = d hx[0] . Wa^T -> gh
      TACO_TRY(tn(w.h[0], c.c2, c.c2, gh2, kCb, kCb, G + c.adapt[0].w, kCb, M, M, 0, s, 1, G + c.adapt[0].b));
      TACO_TRY(launch_conv_gemm(dense_problem(gh2, kCb, PT + t.adapt[0], c.c2, nullptr, gh, c.c2, M, c.c2, kCb, TACO_ACT_NONE), s));
    } else {
      float* tmp = gh; gh = gh2; gh2 = tmp;
    }
    TACO_TRY(hw_group.flush());
  } else if (spk_fused) {
    // multi-speaker encoder, round 6: the four layers' activation-gradient chain THROUGH their adapters is one launch; it leaves
    // d[T|H] (dth) and the gradients of the adapter outputs (dhx) behind, from which every weight gradient and the speaker sites'
    // small chains are formed in grouped launches -- 12 launches where rounds 1-5 issued ~50.
    HighwayStackBwdArgs hb;
    hb.g = gh; hb.gout = gh2; hb.M = M; hb.nl = 4;
    for (int l = 0; l < 4; ++l) {
      hb.wT[l] = PT + t.hw[l]; hb.th[l] = w.th[l]; hb.x[l] = w.hx[l];
      hb.dth[l] = sc.gA + (int64_t)l * M * 2 * kCb;
      hb.waT[l] = PT + t.adapt[l];
      hb.dhx[l] = sc.gA + (int64_t)4 * M * 2 * kCb + (int64_t)l * M * kCb;   // (gA is (M, K*128) = (M, 2048): 1024 + 512 columns used)
    }
    TACO_TRY(launch_highway_stack_bwd(hb, s));
    {
      TnGroup hw_group(s);
      for (int l = 0; l < 4; ++l) {
        TACO_TRY(tn(w.hx[l], kCb, kCb, hb.dth[l], 2 * kCb, kCb, G + c.hwT[l].w, kCb, M, M, 0, s, 1, G + c.hwT[l].b));
        TACO_TRY(tn(w.hx[l], kCb, kCb, hb.dth[l] + kCb, 2 * kCb, kCb, G + c.hwH[l].w, kCb, M, M, 0, s, 1, G + c.hwH[l].b));
        // adapter kernel rows [0,128) (the h part) + its bias: d ba = sum over all rows of d hx[l]
        TACO_TRY(tn(w.h[l], kCb, kCb, hb.dhx[l], kCb, kCb, G + c.adapt[l].w, kCb, M, M, 0, s, 1, G + c.adapt[l].b));
      }
      TACO_TRY(hw_group.flush());
    }
    // per-sequence bias path: d rowb_l[b] = sum_t d hx[l][b,t]; rowb = sv . Wa[128:] + ba; sv = relu(dense(spk)).  Nothing on the
    // activation-gradient chain reads any of it (it ends in parameter gradients and in d spk_e, which only the speaker table's
    // scatter at the end of the pass consumes): the whole chain of small launches goes to the weight-gradient stream.
    hipStream_t q = s;
    TACO_TRY(tn_route(s, &q));
    for (int l = 0; l < 4; ++l) TACO_TRY(launch_colsum_batched(hb.dhx[l], kCb, w.dsm[l], B, T, kCb, q));
    {
      TnGroup g2(q);
      for (int l = 0; l < 4; ++l)
        TACO_TRY(tn(w.sv[l], kCb, kCb, w.dsm[l], kCb, kCb, G + c.adapt[l].w + (int64_t)kCb * kCb, kCb, B, B, 0, q));
      TACO_TRY(g2.flush());
    }
    ConvGemmBatch b3;
    b3.n = 4;
    for (int l = 0; l < 4; ++l)
      b3.p[l] = dense_problem(w.dsm[l], kCb, PT + t.adapt_s[l], kCb, nullptr, w.dsm2[l], kCb, B, kCb, kCb, TACO_ACT_NONE);
    TACO_TRY(launch_conv_gemm_batch(b3, q));
    // ReLU backward of sv[0..3] and h0 in one pass over the contiguous (5,B,128) blocks
    TACO_TRY(launch_act_bwd(w.sv[0], w.dsm2[0], nullptr, w.dsm2[0], (int64_t)5 * B * kCb, TACO_ACT_RELU, q));
    const DenseP* dps[5] = {&c.spkd[0], &c.spkd[1], &c.spkd[2], &c.spkd[3], &c.gru_init};
    const int64_t wTs[5] = {t.spkd[0], t.spkd[1], t.spkd[2], t.spkd[3], t.gru_init};
    {
      TnGroup g3(q);
      for (int i = 0; i < 5; ++i) TACO_TRY(tn(w.spk_e, 16, 16, w.dsm2[i], kCb, kCb, G + dps[i]->w, kCb, B, B, 0, q, 1, G + dps[i]->b));
      TACO_TRY(g3.flush());
    }
    ConvGemmBatch b4;
    b4.n = 5;
    for (int i = 0; i < 5; ++i)
      b4.p[i] = dense_problem(w.dsm2[i], kCb, PT + wTs[i], 16, nullptr, w.dspk_part + (int64_t)i * B * 16, 16, B, 16, kCb, TACO_ACT_NONE);
    TACO_TRY(launch_conv_gemm_batch(b4, q));
    TACO_TRY(launch_colsum_batched(w.dspk_part, B * 16, w.dspk_e, 1, 5, B * 16, q));   // d spk_e = sum of the five sites' parts
    // (gh2 = the gradient of the residual sum `res` = h[0]: nothing in front of the adapters of layer 0)
    float* tmp = gh; gh = gh2; gh2 = tmp;
  } else {
  TnGroup hw_group(s);
  for (int l = 3; l >= 0; --l) {
    float* dth = sc.gA + (int64_t)l * M * 2 * kCb;
    TACO_TRY(launch_highway_combine_bwd(w.th[l], w.hx[l], gh, dth, dxd, M, s));
    TACO_TRY(tn(w.hx[l], kCb, kCb, dth, 2 * kCb, kCb, G + c.hwT[l].w, kCb, M, M, 0, s, 1, G + c.hwT[l].b));
    TACO_TRY(tn(w.hx[l], kCb, kCb, dth + kCb, 2 * kCb, kCb, G + c.hwH[l].w, kCb, M, M, 0, s, 1, G + c.hwH[l].b));
    ConvGemmProblem p = dense_problem(dth, 2 * kCb, PT + t.hw[l], kCb, nullptr, gh2, kCb, M, kCb, 2 * kCb, TACO_ACT_NONE);
    p.residual = dxd;
    p.ldr = kCb;
    TACO_TRY(launch_conv_gemm(p, s));        // gh2 = d hx[l]
    if (!c.has_adapt[l]) {
      float* tmp = gh; gh = gh2; gh2 = tmp;  // hx[l] == h[l]
      continue;
    }
    const int cin_h = c.spk ? kCb : c.c2;
    // adapter weight (rows of h) + bias, and d h[l] = d hx[l] . Wa[:cin_h]^T  -> gh (old gh is dead)
    TACO_TRY(tn(w.h[l], cin_h, cin_h, gh2, kCb, kCb, G + c.adapt[l].w, kCb, M, M, 0, s, 1, G + c.adapt[l].b));
    TACO_TRY(launch_conv_gemm(dense_problem(gh2, kCb, PT + t.adapt[l], cin_h, nullptr, gh, cin_h, M, cin_h, kCb, TACO_ACT_NONE), s));
    if (c.spk) {
      // per-sequence bias path: d rowb[b] = sum_t d hx[l][b,t]; rowb = sv . Wa[128:] + ba; sv = relu(dense(spk))
      TACO_TRY(launch_colsum_batched(gh2, kCb, w.dsmall, B, T, kCb, s));
      TACO_TRY(tn(w.sv[l], kCb, kCb, w.dsmall, kCb, kCb, G + c.adapt[l].w + (int64_t)kCb * kCb, kCb, B, B, 0, s));
      TACO_TRY(launch_conv_gemm(dense_problem(w.dsmall, kCb, PT + t.adapt_s[l], kCb, nullptr, w.dsmall2, kCb, B, kCb, kCb,
                                              TACO_ACT_NONE), s));
      TACO_TRY(launch_act_bwd(w.sv[l], w.dsmall2, nullptr, w.dsmall2, (int64_t)B * kCb, TACO_ACT_RELU, s));
      TACO_TRY(spk_dense_bwd(w.dsmall2, c.spkd[l], t.spkd[l]));
    }
    TACO_TRY(hw_group.flush());
    hw_group.batch.n = 0;
  }
  TACO_TRY(hw_group.flush());
  }   // (group scope: later weight gradients launch on their own again)
  float* dres = gh;      // (M, c2) gradient wrt `res`
  // ---- res = bn(conv(pj1)) + x ----
  float* dz2 = sc.gG;    // (M,c2)
  TACO_TRY(launch_affine_act_bwd(w.pj2pre, P + c.p2_g, dres, dz2, G + c.p2_g, G + c.p2_be, M, c.c2, TACO_ACT_NONE, s));
  TACO_TRY(tn(w.pj1, c.c1, c.c1, dz2, c.c2, c.c2, G + c.p2_w, c.c2, M, T, 1, s, 3, G + c.p2_b));
  float* dpj1 = sc.alt_dpj1 ? sc.alt_dpj1 : sc.gD;   // (M,c1)
  {
    ConvGemmProblem p;
    p.A = dz2; p.lda = c.c2; p.W = PT + t.p2; p.ldw = c.c1; p.C = dpj1; p.ldc = c.c1; p.M = M; p.N = c.c1; p.K = c.c2;
    p.taps = 3; p.T = T; p.pad_l = 1; p.act = TACO_ACT_NONE;
    TACO_TRY(launch_conv_gemm(p, s));
  }
  float* dz1 = sc.alt_dz1 ? sc.alt_dz1 : sc.gC;    // (M,c1)  (dxg no longer needed unless its weight gradients run on the side stream)
  TACO_TRY(launch_affine_act_bwd(w.pj1pre, P + c.p1_g, dpj1, dz1, G + c.p1_g, G + c.p1_be, M, c.c1, TACO_ACT_RELU, s));
  TACO_TRY(tn(w.pool, KC, KC, dz1, c.c1, c.c1, G + c.p1_w, c.c1, M, T, 1, s, 3, G + c.p1_b));
  if (seg_after_p1 >= 0) {
    // every gradient in [p1_w, end of this CBHG) has been enqueued: on the weight-gradient stream (ordered behind the main
    // stream's work so far by tn_route's event), or on `s` itself
    hipStream_t q;
    TACO_TRY(tn_route(s, &q));
    TACO_TRY(record_segment(seg_after_p1, q));
  }
  float* dpool = sc.alt_dpool ? sc.alt_dpool : sc.gA;  // (M,KC)
  float* dbank = sc.gB;  // (M,KC)
  {
    // d pool = dz1 (*) W_p1^T, then back through max-pool, BN-affine and the bank's ReLU.  Where the DMA kernel takes the GEMM
    // (and no fixed summation order is asked for) all of it is the GEMM's epilogue: d pool is never written (gemm2.hip, pool == 2)
    ConvGemmProblem p;
    p.A = dz1; p.lda = c.c1; p.W = PT + t.p1; p.ldw = KC; p.C = dpool; p.ldc = KC; p.M = M; p.N = KC; p.K = c.c1;
    p.taps = 3; p.T = T; p.pad_l = 1; p.act = TACO_ACT_NONE;
    int rc = TACO_ENOTFOUND;
    const char* nf = getenv("TACO_NO_POOL_FUSE");
    if (!taco_deterministic() && !(nf && atoi(nf) != 0)) {
      ConvGemmProblem q = p;
      q.C = dbank; q.pool = 2; q.pool_x = w.bank; q.scale = P + c.bank_g; q.shift = P + c.bank_be;
      q.scale_mul = 1.0f / sqrtf(1.0f + kBnEps); q.pool_dgamma = G + c.bank_g; q.pool_dbeta = G + c.bank_be;
      rc = launch_conv_gemm(q, s);
    }
    if (rc == TACO_ENOTFOUND) {
      TACO_TRY(launch_conv_gemm(p, s));
      // (d relu included: dbank is the gradient of the bank's pre-activation)
      TACO_TRY(launch_bn_maxpool_bwd(w.bank, P + c.bank_g, P + c.bank_be, dpool, dbank, G + c.bank_g, G + c.bank_be, B, T, KC, s));
    } else {
      TACO_TRY(rc);
    }
  }
  // ---- conv bank: weight/bias grads per width; input grad = residual path + sum over widths, accumulated with fp32
  //      atomics by ONE batched launch (all K transposed convolutions run concurrently instead of as a dependent chain) ----
  {
    if (!sc.dx_zeroed) {
      hipError_t e = hipMemsetAsync(dx_out, 0, (size_t)M * c.cin * sizeof(float), s);
      taco_tail_touch(s);
      if (e != hipSuccess) {
        taco_set_error("cbhg_bwd: memset: %s", hipGetErrorString(e));
        return TACO_ELAUNCH;
      }
    }
    ConvGemmBatch batch;
    batch.n = c.K;
    TnGroup bank_group(s);   // K independent conv-bank weight gradients: one grid
    for (int k = 1; k <= c.K; ++k) {
      const float* dzk = dbank + (k - 1) * kCb;
      TACO_TRY(tn(x, c.cin, c.cin, dzk, KC, kCb, G + c.bank_w[k - 1], kCb, M, T, (k - 1) / 2, s, k, G + c.bank_b[k - 1]));
      ConvGemmProblem& p = batch.p[c.K - k];
      p = ConvGemmProblem();
      p.A = dzk; p.lda = KC; p.W = PT + t.bank[k - 1]; p.ldw = c.cin; p.C = dx_out; p.ldc = c.cin; p.M = M; p.N = c.cin;
      p.K = kCb; p.taps = k; p.T = T; p.pad_l = (k - 1) - (k - 1) / 2; p.act = TACO_ACT_NONE; p.atomic_out = 1;
      if (k == 1) {   // the residual connection (d res / d x = identity, c2 == cin) rides on the cheapest problem
        p.residual = dres;
        p.ldr = c.c2;
      }
    }
    TACO_TRY(bank_group.flush());
    if (seg_after_bank >= 0) {
      hipStream_t q;
      TACO_TRY(tn_route(s, &q));
      TACO_TRY(record_segment(seg_after_bank, q));
    }
    if (taco_deterministic()) {
      // fixed order: the K transposed convolutions run one after the other, each adding to dx_out through the residual input
      for (int i = c.K - 1; i >= 0; --i) {   // batch.p[K - 1] is the k = 1 problem carrying the dres residual
        ConvGemmProblem q = batch.p[i];
        q.atomic_out = 0;
        if (i != c.K - 1) { q.residual = dx_out; q.ldr = c.cin; }
        TACO_TRY(launch_conv_gemm(q, s));
      }
    } else {
      // Default: ONE reduction over the taps of all K widths (ConvGemmProblem::bank_filters), cut into S equal chunks of the
      // (tap, k-tile) sequence = S x m-tiles workgroups that fill the chip once, equal work each.  Every chunk writes its
      // partial tile to its own slab (the forward pass's proj1 slabs are free here) and a second pass sums the slabs in
      // order and adds the residual: no atomics -- 13 M / 8 M of them cost 80 / 35 us of these two launches
      // (profiles/r04_bank_gather.txt) -- and a deterministic result.  Falls back to the K-problem atomic batch when the
      // transposed kernels are not contiguous, the slabs do not fit or the DMA kernel does not take the shape
      // (TACO_NO_BANK_GATHER=1: always).
      bool contiguous = true;
      for (int k = 2; k <= c.K; ++k) contiguous = contiguous && t.bank[k - 1] == t.bank[k - 2] + (int64_t)(k - 1) * kCb * c.cin;
      int rc = TACO_ENOTFOUND;
      const int64_t mn = (int64_t)M * c.cin;
      if (contiguous && w.tapsplit && c.cin % 4 == 0 && getenv("TACO_NO_BANK_GATHER") == nullptr) {
        const int taps = c.K * (c.K + 1) / 2, nit = taps * (kCb / 32);
        const int mtiles = cdiv(M, 128) * cdiv(c.cin, 128);
        int64_t S = 512 / mtiles;
        S = std::min<int64_t>(std::min<int64_t>(S, kMaxGemmBatch), w.tapsplit_floats / mn);
        if (S >= 2) {
          const int per = cdiv(nit, (int)S);
          ConvGemmBatch g;
          g.n = cdiv(nit, per);
          for (int ch = 0; ch < g.n; ++ch) {
            ConvGemmProblem& p = g.p[ch];
            p = ConvGemmProblem();
            p.A = dbank; p.lda = KC; p.W = PT + t.bank[0]; p.ldw = c.cin; p.C = w.tapsplit + ch * mn; p.ldc = c.cin; p.M = M;
            p.N = c.cin; p.K = kCb; p.taps = taps; p.T = T; p.pad_l = (c.K - 1) - (c.K - 1) / 2; p.act = TACO_ACT_NONE;
            p.bank_filters = c.K;
            p.it0 = ch * per;
            p.it1 = std::min(nit, (ch + 1) * per);
            conv_gemm_set_flags(p);
          }
          const int pslot = taco_prof_begin(2, s);
          taco_prof_label(2, pslot, "nn bank-gather F=%d S=%d M=%d N=%d K=%d taps=%d + slab sum", c.K, g.n, M, c.cin, kCb, taps);
          rc = launch_conv_gemm2(g, s, /*force=*/true);
          if (rc == TACO_OK) {
            ConvGemmProblem fin;
            fin.M = M; fin.N = c.cin; fin.T = T; fin.C = dx_out; fin.ldc = c.cin; fin.act = TACO_ACT_NONE;
            fin.residual = dres; fin.ldr = c.c2;   // the residual connection (d res / d x = identity, c2 == cin)
            fin.atomic_out = 1;                    // dx_out += ...: zeroed above or pre-loaded by the caller with another term
            rc = launch_conv_gemm_slab_sum(fin, w.tapsplit, g.n, s);
            taco_prof_end(2, pslot, s, 2.0 * M * c.cin * kCb * taps);
          }
        }
      }
      if (rc == TACO_ENOTFOUND) rc = launch_conv_gemm_batch(batch, s);
      TACO_TRY(rc);
    }
  }
  return TACO_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int64_t taco_param_count(const TacoShape* shape) {
  if (validate_shape(shape) != TACO_OK) return TACO_EINVAL;
  return layouts_for(*shape).P.total;
}

static int copy_rows(const std::vector<TacoTensorInfo>& src, TacoTensorInfo* rows, int cap) {
  const int n = (int)src.size();
  if (rows)
    for (int i = 0; i < n && i < cap; ++i) rows[i] = src[i];
  return n;
}

extern "C" int taco_param_table(const TacoShape* shape, TacoTensorInfo* rows, int cap) {
  TACO_TRY(validate_shape(shape));
  return copy_rows(layouts_for(*shape).P.rows, rows, cap);
}

extern "C" int64_t taco_workspace_bytes(const TacoShape* shape, int train) {
  if (validate_shape(shape) != TACO_OK) return TACO_EINVAL;
  const Layouts& L = layouts_for(*shape);
  return (train ? L.Wtrain.total : L.Winfer.total) * (int64_t)sizeof(float);
}

extern "C" int taco_workspace_table(const TacoShape* shape, int train, TacoTensorInfo* rows, int cap) {
  TACO_TRY(validate_shape(shape));
  const Layouts& L = layouts_for(*shape);
  return copy_rows(train ? L.Wtrain.rows : L.Winfer.rows, rows, cap);
}

extern "C" int taco_forward(const TacoShape* shape, const float* params, const int32_t* text, const int32_t* text_length,
                            const int32_t* speaker, const float* mel, const float* stft, const uint8_t* enc_keep1, const uint8_t* enc_keep2,
                            const uint8_t* dec_keep1, const uint8_t* dec_keep2, const uint8_t* sample,
                            float* seq2seq_output, float* output, float* alignments, float* loss, void* workspace,
                            void* stream) {
  TACO_TRY(validate_shape(shape));
  TACO_REQUIRE(params && text && text_length && mel && stft && seq2seq_output && output && alignments && loss && workspace,
               "taco_forward: null pointer argument");
  const Layouts& L = layouts_for(*shape);
  const WsLayout& W = L.Wtrain;
  float* ws = static_cast<float*>(workspace);
  hipStream_t s = as_stream(stream);
  TailScope tails(s, 0, *shape);
  TACO_TRY(forward_impl(*shape, L, W, params, text, text_length, speaker, mel, enc_keep1, enc_keep2, dec_keep1, dec_keep2, sample,
                        seq2seq_output, output, alignments, ws, true, s));
  // add_loss_op (tacotron.py:156-165) + sign gradients for the backward pass
  const int R80 = kMel * shape->r;
  const int64_t MD = (int64_t)shape->B * shape->Td, M2 = MD * shape->r;
  // (the seq2seq term and the zeroing of the loss slots were issued by forward_impl beside the post-net)
  (void)R80; (void)MD;
  TACO_TRY(launch_l1(output, stft, ws + W.dout_pad, 1028, ws + W.loss + 4 + kLossParts, M2, kFft, s));
  return launch_finish_loss(ws + W.loss, ws + W.loss + 4, loss, s);
}

extern "C" int taco_infer(const TacoShape* shape, const float* params, const int32_t* text, const int32_t* text_length,
                          const int32_t* speaker, float* seq2seq_output, float* output, float* alignments, void* workspace, void* stream) {
  TACO_TRY(validate_shape(shape));
  TACO_REQUIRE(params && text && text_length && seq2seq_output && output && alignments && workspace,
               "taco_infer: null pointer argument");
  const Layouts& L = layouts_for(*shape);
  TailScope tails(as_stream(stream), 1, *shape);
  return forward_impl(*shape, L, L.Winfer, params, text, text_length, speaker, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      seq2seq_output, output, alignments, static_cast<float*>(workspace), false, as_stream(stream));
}


extern "C" int taco_backward(const TacoShape* shape, const float* params, const int32_t* text, const int32_t* text_length,
                             const int32_t* speaker, const float* seq2seq_output, const float* alignments, const uint8_t* enc_keep1,
                             const uint8_t* enc_keep2, const uint8_t* dec_keep1, const uint8_t* dec_keep2,
                             const uint8_t* sample, float* grads, void* workspace, void* stream) {
  TACO_TRY(validate_shape(shape));
  TACO_REQUIRE(params && text && text_length && seq2seq_output && alignments && grads && workspace,
               "taco_backward: null pointer argument");
  const Layouts& L = layouts_for(*shape);
  const ParamLayout& PL = L.P;
  const TransLayout& TL = L.T;
  const WsLayout& W = L.Wtrain;
  float* ws = static_cast<float*>(workspace);
  float* G = grads;
  const float* P = params;
  hipStream_t s = as_stream(stream);
  const int B = shape->B, Tt = shape->Tt, Td = shape->Td, r = shape->r, R80 = kMel * r;
  const int M1 = B * Tt, MD = B * Td, M2 = MD * r, F = Td * r;
  float* PT = ws + W.paramsT;

  g_tn_side = nullptr;
  TailScope tails(s, 2, *shape);
  // the weight-image table of this call (the images themselves were built by the taco_forward that ran on this workspace)
  weight_images_clear();
  TACO_TRY(register_weight_images(L, W, P, ws, true, 0, false));
  TACO_TRY(register_weight_images(L, W, P, ws, true, 1, false));
  // ONE batched init launch for every accumulator of the pass: the gradient buffer, [d keys | E] (one (M1, 512) buffer), the small
  // decoder weight-gradient factors, the two CBHG input-gradient accumulators (when they have buffers of their own) and the
  // decoder exchange area (the forward kernel is long done with it)
  const bool own_dx = getenv("TACO_NO_SIDE_TN") == nullptr && !side_stream().off &&
                      side_stream().side != nullptr;
  {
    InitBatch ib;
    TACO_TRY(ib.fill(G, PL.total));
    TACO_TRY(ib.fill(ws + W.dkeys, (int64_t)M1 * 2 * kAtt));
    TACO_TRY(ib.fill(ws + W.bc_g, W.bc_cp + kPre1 - W.bc_g));
    TACO_TRY(ib.fill(ws + W.xchg, decoder_xchg_bytes(B, Tt) / 4));
    if (own_dx) {
      // the post-net's input-gradient accumulator starts from the direct L1 term sign(s2s - mel) (left by the forward pass)
      // instead of from zero: d seq2seq_output is complete when the K-way bank backward has added its part -- no add pass.
      // (Deterministic mode overwrites the accumulator with a fixed-order chain and keeps the explicit add.)
      if (!taco_deterministic()) TACO_TRY(ib.copy(ws + W.post_dx, ws + W.ds2s, (int64_t)M2 * kMel));
      else TACO_TRY(ib.fill(ws + W.post_dx, (int64_t)M2 * kMel));
      if (!PL.enc.spk || spk_fused_form(PL.enc)) TACO_TRY(ib.fill(ws + W.enc_dx, (int64_t)M1 * kCb));
    }
    TACO_TRY(launch_init_batch(ib, s));
  }
  // (the transposed weights PT and the transposed decoder composites were built by taco_forward on this workspace)
  BwdScratch sc{ws + W.gA, ws + W.gB, ws + W.gC, ws + W.gD, ws + W.gE, ws + W.gF, ws + W.gG};
  CbhgBufs pb = cbhg_bufs(ws, W.post), eb = cbhg_bufs(ws, W.enc);

  // ---- final dense (tacotron.py:148): output = post_out . Wd + bd ----
  const float* dOutPad = ws + W.dout_pad;  // (M2, 1028) = sign(output - stft), written by taco_forward
  float* dPostOut = sc.gG;  // (M2,256); consumed by the bi-GRU backward before gG is reused
  TACO_TRY(launch_conv_gemm(dense_problem(dOutPad, 1028, PT + TL.post_dense, 2 * kCb, nullptr, dPostOut, 2 * kCb, M2, 2 * kCb,
                                          1028 /* K padded: dOutPad pad columns and WdT pad rows are zero */, TACO_ACT_NONE), s));
  // (weight gradient on the side stream, forked here: it runs beside the post-net bi-GRU backward -- the first kernel of
  // cbhg_bwd -- which occupies 64 of 256 CUs)
  hipStream_t side = side_fork(s);
  TACO_TRY(tn(pb.out, 2 * kCb, 2 * kCb, dOutPad, 1028, kFft, G + PL.post_dense.w, kFft, M2, M2, 0, side, 1, G + PL.post_dense.b, 1028));
  // ---- post-net CBHG (input = seq2seq_output viewed as (B, Td*r, 80)) ----
  // The CBHG's ~27 weight-gradient GEMMs feed nothing but the gradient buffer: they go to the side stream (g_tn_side) and run
  // beside the activation-gradient chain, much of which is small launches that leave most CUs idle.  (Running them UNDERNEATH
  // the BPTT kernel instead was measured in rounds 2 and 3 -- both sides run ~1.5x slower while they share CUs, net 0.6 % -- and
  // removed in round 4 together with the data-parallel mode that overlapped collectives with that launch.)
  BwdScratch scp = sc;
  float* dPostIn = sc.gC;   // (M2, 80)
  const bool side_tn = side != s && getenv("TACO_NO_SIDE_TN") == nullptr;
  if (side_tn) {
    scp.alt_dpj1 = ws + W.post_dpj1; scp.alt_dz1 = ws + W.post_dz1; scp.alt_dpool = ws + W.post_dpool;
    dPostIn = ws + W.post_dx;
    scp.dx_zeroed = own_dx;
    g_tn_side = side;
  }
  pb.tapsplit = ws + W.tapsplit;   // (free in the backward pass: the conv-bank input gradient's partial tiles)
  pb.tapsplit_floats = W.tapsplit_floats;
  const int rc_post = cbhg_bwd(P, PT, G, PL.post, TL.post, seq2seq_output, dPostOut, B, F, pb, scp, dPostIn, -1, -1, s);
  g_tn_side = nullptr;
  TACO_TRY(rc_post);
  // d seq2seq_output = sign(s2s - mel) + post-net path
  float* dS2S = ws + W.ds2s_tot;
  if (side_tn && own_dx && !taco_deterministic()) dS2S = dPostIn;   // (accumulated on top of the L1 term, see the init launch)
  else TACO_TRY(launch_add(ws + W.ds2s, dPostIn, dS2S, (int64_t)MD * R80, s));
  TACO_TRY(side_join(s, side));
  // Gradient segment 4 (post-net CBHG + final dense) is final HERE, but it is ANNOUNCED only after the BPTT kernel below: that
  // kernel is a persistent launch that needs all 256 workgroups co-resident, one per CU, and a collective's kernel that took
  // CUs first would leave part of every cluster spinning on peers that cannot start.  The bytes travel under the encoder
  // backward instead -- 1.5 ms of ordinary kernels.

  // ---- decoder BPTT ----
  float* gs = ws + W.gstash;
  const float* st = ws + W.stash;
  {
    DecBwdArgs a;
    DecWeights& w = a.wT;
    w.pre_w1 = PT + TL.dec_pre1; w.pre_w2 = PT + TL.dec_pre2; w.in_w = PT + TL.in_proj;
    w.pre_b1 = w.pre_b2 = w.in_b = w.out_b = nullptr;
    for (int l = 0; l < 3; ++l) {
      w.gw[l] = PT + TL.gw[l]; w.cw[l] = PT + TL.cw[l]; w.gb[l] = w.cb[l] = nullptr;
    }
    w.out_w = PT + TL.out_proj; w.q_w = PT + TL.q_w; w.att_w = PT + TL.att_w; w.att_v = P + PL.att_v;
    a.att_v = P + PL.att_v;
    a.wot = ws + W.bc_wot; a.wdx = ws + W.bc_wdx;
    a.keys = ws + W.keys; a.vwx = ws + W.vwxc; a.text_length = text_length;
    a.keep1 = dec_keep1; a.keep2 = dec_keep2; a.sample = sample;
    a.dout = dS2S; a.out = seq2seq_output; a.align = alignments; a.stash = st; a.gstash = gs;
    a.dkeys = ws + W.dkeys; a.ldk = 2 * kAtt; a.datt_v = ws + W.dattv;
    a.hoisted = 1;
    a.xchg = ws + W.xchg; a.err = reinterpret_cast<int*>(ws + W.err) + 1;
    a.trace = getenv("TACO_DEC_TRACE") ? reinterpret_cast<long long*>(ws + W.err + 16) + 128 : nullptr;
    a.B = B; a.Tt = Tt; a.Td = Td; a.r = r; a.P = 1;
    a.xchg_zeroed = 1;
    const int slot = prof_begin(1, s);
    int rc = launch_decoder3_bwd(a, s);
    if (rc == TACO_ENOTFOUND) rc = launch_decoder_bwd(a, s);
    TACO_TRY(rc);
    prof_end(1, slot, s);
    TACO_TRY(record_segment(4, s));
  }
  // ---- attention memory.  The kernel never forms the context or its gradient (decoder.hip); everything they carried follows
  //      from E[b] = sum_t alignments[b,t-1]^T dx[b,t]  (Tt x 256 per row; one batched launch):
  //        d values = E Wx_c^T ,   sum_t ctx_{t-1}^T dx_t = sum_b values[b]^T E[b]   (the context rows of Gx below)
  //      keys = values . Wm ----
  {
    GemmTnArgs a;
    a.A = alignments; a.lda = Tt; a.Y = gs + kGsX; a.ldy = kGsRec; a.W = ws + W.dvalues; a.ldw = 2 * kAtt;
    a.M = Td; a.N = kDec; a.K = Tt; a.taps = 1; a.T = Td; a.pad_l = 1; a.batch = B;
    a.strideA = (int64_t)Td * Tt; a.strideY = (int64_t)Td * kGsRec; a.strideW = (int64_t)Tt * 2 * kAtt;
    TACO_TRY(launch_gemm_tn(a, false, s));
  }
  float* dValTot = sc.gE;   // (M1,256) = d keys Wm^T + E Wx_c^T
  TACO_TRY(launch_conv_gemm(dense_problem(ws + W.dkeys, 2 * kAtt, ws + W.bc_wmx, 2 * kCb, nullptr, dValTot, 2 * kCb, M1, 2 * kCb,
                                          2 * kAtt, TACO_ACT_NONE), s));
  float* dEnc = sc.gG;      // (M1,256)
  TACO_TRY(launch_mask_rows(dValTot, text_length, dEnc, B, Tt, 2 * kCb, s));
  // ---- decoder weight gradients: dense GEMMs over the B*Td stashed rows, all independent -> one grouped launch, on the
  //      side stream, forked HERE so that they start together with the encoder bi-GRU backward (first kernel of cbhg_bwd),
  //      which occupies only 64 CUs; nothing below depends on them ----
  {
    hipStream_t main_s = s;
    hipStream_t s = side_fork(main_s);   // shadows the main stream inside this block
    // total d cell_output_t = direct part + dx_{t+1} Wx_o^T  (the kernel carries the second term straight into d(x + h3); the
    // sum is what the output projection's weight gradient below needs): one GEMM over the B*Td rows, A row t+1 against row t
    {
      ConvGemmProblem p = dense_problem(gs + kGsX, kGsRec, ws + W.bc_fa, dec_fan_cols(r), nullptr, gs + kGsO, kGsRec, MD, R80, kDec,
                                        TACO_ACT_NONE);
      p.T = Td; p.pad_l = -1;
      p.residual = dS2S; p.ldr = R80;
      TACO_TRY(launch_conv_gemm(p, s));
    }
    // d attention_v = sum over batch rows of the kernel's per-row partials, in row order (no atomics)
    TACO_TRY(launch_colsum_batched(ws + W.dattv, kAtt, G + PL.att_v, 1, B, kAtt, s));
    // decoder pre_net backward for ALL steps (the kernel only ran it where the recurrence needed it, i.e. into steps fed by the
    // previous output): d p2pre = [p2 > 0] keep2 (dx Wi_p^T), d p1pre = [p1 > 0] keep1 (d p2pre W2^T) -- two GEMMs over the B*Td rows
    {
      ConvGemmProblem q2 = dense_problem(gs + kGsX, kGsRec, PT + TL.in_proj, kPre2 + kAtt, nullptr, gs + kGsP2, kGsRec, MD, kPre2,
                                         kDec, TACO_ACT_NONE);
      TACO_TRY(launch_conv_gemm(q2, s));
      TACO_TRY(launch_mask_pos(gs + kGsP2, kGsRec, st + kStP2, kStRec, MD, kPre2, dec_keep2 ? 2.f : 1.f, s));
      ConvGemmProblem q1 = dense_problem(gs + kGsP2, kGsRec, PT + TL.dec_pre2, kPre1, nullptr, gs + kGsP1, kGsRec, MD, kPre1, kPre2,
                                         TACO_ACT_NONE);
      TACO_TRY(launch_conv_gemm(q1, s));
      TACO_TRY(launch_mask_pos(gs + kGsP1, kGsRec, st + kStP1, kStRec, MD, kPre1, dec_keep1 ? 2.f : 1.f, s));
    }
    TnGroup dec_group(s);
    TACO_TRY(tn(ws + W.values, kAtt, 2 * kCb, ws + W.dkeys, 2 * kAtt, kAtt, G + PL.mem_w, kAtt, M1, M1, 0, s));   // d Wm = values^T d keys
    const float* prein = ws + W.prein;
    TACO_TRY(tn(prein, kMel, kMel, gs + kGsP1, kGsRec, kPre1, G + PL.dec_pre1.w, kPre1, MD, Td, 0, s, 1, G + PL.dec_pre1.b));
    TACO_TRY(tn(st + kStP1, kStRec, kPre1, gs + kGsP2, kGsRec, kPre2, G + PL.dec_pre2.w, kPre2, MD, Td, 0, s, 1, G + PL.dec_pre2.b));
    // in-proj rows [0,128): pre-net output of step t.  Rows [128,384) (attention of step t-1) and the attention layer follow
    // from Gx = sum_t [cell_output ; context]_{t-1}^T dx_t below: the kernel never forms the attention vector or its gradient.
    TACO_TRY(tn(st + kStP2, kStRec, kPre2, gs + kGsX, kGsRec, kDec, G + PL.in_proj.w, kDec, MD, Td, 0, s, 1, G + PL.in_proj.b));
    TACO_TRY(tn(seq2seq_output, R80, R80, gs + kGsX, kGsRec, kDec, ws + W.bc_g, kDec, MD, Td, 1, s));
    TACO_TRY(tn(ws + W.values, kAtt, kAtt, ws + W.dvalues, 2 * kAtt, kDec, ws + W.bc_g + (int64_t)R80 * kDec, kDec, M1, M1, 0, s));
    for (int l = 0; l < 3; ++l) {
      const float* inp = l == 0 ? st + kStX : st + kStH + (l - 1) * kDec;
      const float* dG = gs + kGsG + l * 512;
      const float* dC = gs + kGsC + l * kDec;
      TACO_TRY(tn(inp, kStRec, kDec, dG, kGsRec, 2 * kDec, G + PL.gru[l].wg, 2 * kDec, MD, Td, 0, s, 1, G + PL.gru[l].bg));
      TACO_TRY(tn(st + kStH + l * kDec, kStRec, kDec, dG, kGsRec, 2 * kDec, G + PL.gru[l].wg + (int64_t)kDec * 2 * kDec, 2 * kDec,
                  MD, Td, 1, s));
      TACO_TRY(tn(inp, kStRec, kDec, dC, kGsRec, kDec, G + PL.gru[l].wc, kDec, MD, Td, 0, s, 1, G + PL.gru[l].bc));
      TACO_TRY(tn(st + kStRH + l * kDec, kStRec, kDec, dC, kGsRec, kDec, G + PL.gru[l].wc + (int64_t)kDec * kDec, kDec, MD, Td, 0, s));
    }
    // out-proj: d cell_output(total) = direct + dq Wq^T + (sampled) dp1s W1^T on the last frame; the kernel stashes only the
    // direct part, the rest follows from H1 = sum_t y_t^T dq_t, H2 = sum_t y_t^T dp1s_t  (y = x + h3)
    TACO_TRY(tn(st + kStY, kStRec, kDec, gs + kGsO, kGsRec, R80, G + PL.out_proj.w, R80, MD, Td, 0, s, 1, G + PL.out_proj.b));
    TACO_TRY(tn(st + kStY, kStRec, kDec, gs + kGsQ, kGsRec, kAtt, ws + W.bc_h1, kAtt, MD, Td, 0, s, 1, ws + W.bc_cq));
    TACO_TRY(tn(st + kStY, kStRec, kDec, gs + kGsP1S, kGsRec, kPre1, ws + W.bc_h2, kPre1, MD, Td, 0, s, 1, ws + W.bc_cp));
    TACO_TRY(tn(seq2seq_output, R80, R80, gs + kGsQ, kGsRec, kAtt, G + PL.q_w, kAtt, MD, Td, 0, s));
    TACO_TRY(dec_group.flush());
    const float* Gx = ws + W.bc_g;
    ConvGemmBatch b1;
    b1.n = 4;
    // d att_w = Gx Wi_a^T
    b1.p[0] = dense_problem(Gx, kDec, PT + TL.in_proj + kPre2, kPre2 + kAtt, nullptr, G + PL.att_w, kAtt, R80 + kAtt, kAtt, kDec,
                            TACO_ACT_NONE);
    // d in_proj rows [128,384) = Wa^T Gx
    b1.p[1] = dense_problem(PT + TL.att_w, R80 + kAtt, Gx, kDec, nullptr, G + PL.in_proj.w + (int64_t)kPre2 * kDec, kDec, kAtt, kDec,
                            R80 + kAtt, TACO_ACT_NONE);
    // d out_proj += H1 Wq^T ;  d out_proj bias += (sum dq) Wq^T
    b1.p[2] = dense_problem(ws + W.bc_h1, kAtt, PT + TL.q_w, R80, nullptr, G + PL.out_proj.w, R80, kDec, R80, kAtt, TACO_ACT_NONE);
    b1.p[2].residual = G + PL.out_proj.w; b1.p[2].ldr = R80;
    b1.p[3] = dense_problem(ws + W.bc_cq, kAtt, PT + TL.q_w, R80, nullptr, G + PL.out_proj.b, R80, 1, R80, kAtt, TACO_ACT_NONE);
    b1.p[3].residual = G + PL.out_proj.b; b1.p[3].ldr = R80;
    TACO_TRY(launch_conv_gemm_batch(b1, s));
    ConvGemmBatch b2;
    b2.n = 2;
    // last-frame columns: += H2 W1^T ; bias += (sum dp1s) W1^T
    b2.p[0] = dense_problem(ws + W.bc_h2, kPre1, PT + TL.dec_pre1, kMel, nullptr, G + PL.out_proj.w + (R80 - kMel), R80, kDec, kMel,
                            kPre1, TACO_ACT_NONE);
    b2.p[0].residual = G + PL.out_proj.w + (R80 - kMel); b2.p[0].ldr = R80;
    b2.p[1] = dense_problem(ws + W.bc_cp, kPre1, PT + TL.dec_pre1, kMel, nullptr, G + PL.out_proj.b + (R80 - kMel), R80, 1, kMel,
                            kPre1, TACO_ACT_NONE);
    b2.p[1].residual = G + PL.out_proj.b + (R80 - kMel); b2.p[1].ldr = R80;
    TACO_TRY(launch_conv_gemm_batch(b2, s));
    // gradient segment 3 (memory layer + decoder) is final here: everything the main stream contributed (memory-layer
    // kernel, attention_v from the BPTT kernel) was enqueued before this side stream forked
    TACO_TRY(record_segment(3, s));
  }
  // ---- encoder CBHG ----
  float* dP2 = sc.gC;       // (M1,128)
  if (PL.enc.spk) {
    TACO_REQUIRE(speaker != nullptr, "num_speakers=%d but no speaker ids were given", shape->S);
    eb.spk_e = ws + W.spk_e;
    eb.dspk_e = ws + W.dspk_e;
  }
  BwdScratch sce = sc;
  float *pre_dz2 = sc.gD, *pre_dz1 = sc.gE, *pre_demb = sc.gF;
  // (the per-layer speaker form, TACO_SPK_UNFUSED=1, keeps its weight gradients on the main stream: their operands live in ping-pong buffers)
  if (side_tn && side_stream().side && (!PL.enc.spk || spk_fused_form(PL.enc))) {
    // (the side stream is still busy with the decoder weight gradients forked above; the encoder's queue up behind them)
    sce.alt_dpj1 = ws + W.enc_dpj1; sce.alt_dz1 = ws + W.enc_dz1; sce.alt_dpool = ws + W.enc_dpool;
    dP2 = ws + W.enc_dx;
    sce.dx_zeroed = own_dx;
    pre_dz2 = ws + W.pre_dz2; pre_dz1 = ws + W.pre_dz1; pre_demb = ws + W.pre_demb;
    g_tn_side = side_stream().side;
  }
  eb.tapsplit = ws + W.tapsplit;
  eb.tapsplit_floats = W.tapsplit_floats;
  int rc_enc = cbhg_bwd(P, PT, G, PL.enc, TL.enc, ws + W.p2, dEnc, B, Tt, eb, sce, dP2, 2, 1, s);
  if (rc_enc != TACO_OK) {
    g_tn_side = nullptr;
    return rc_enc;
  }
  if (PL.enc.spk) {   // (fused form: d spk_e was summed on the weight-gradient stream -- the speaker table's scatter follows it there)
    hipStream_t q = s;
    if (spk_fused_form(PL.enc)) TACO_TRY(tn_route(s, &q));
    TACO_TRY(launch_embedding_bwd(ws + W.dspk_e, speaker, G + PL.spk_embed, B, shape->S, q, 16));
  }
  // ---- encoder pre_net + embedding ----
  float* dz2 = pre_dz2;
  float* dz1 = pre_dz1;
  float* dEmb = pre_demb;
  if (!getenv("TACO_NO_PRENET_FUSE")) {
    // the activation-gradient chain d p2 -> dz2 -> dz1 -> d embedding in one launch (prenet.hip); the two weight gradients follow
    PrenetArgs pa;
    pa.x = dP2; pa.x_out = dz2; pa.y2_in = ws + W.p2; pa.y1_in = ws + W.p1; pa.keep2 = enc_keep2; pa.keep1 = enc_keep1;
    pa.w1 = PT + TL.enc_pre2; pa.w2 = PT + TL.enc_pre1; pa.y1 = dz1; pa.y2 = dEmb; pa.M = M1;
    TACO_TRY(launch_prenet_bwd(pa, s));
    TnGroup pre_group(s);   // the two weight gradients as one grouped launch: they are the tail of the step
    TACO_TRY(tn(ws + W.p1, kPre1, kPre1, dz2, kPre2, kPre2, G + PL.enc_pre2.w, kPre2, M1, M1, 0, s, 1, G + PL.enc_pre2.b));
    TACO_TRY(tn(ws + W.emb, kEmbed, kEmbed, dz1, kPre1, kPre1, G + PL.enc_pre1.w, kPre1, M1, M1, 0, s, 1, G + PL.enc_pre1.b));
    TACO_TRY(pre_group.flush());
  } else {
    TACO_TRY(launch_act_bwd(ws + W.p2, dP2, enc_keep2, dz2, (int64_t)M1 * kPre2, TACO_ACT_RELU, s));
    TACO_TRY(tn(ws + W.p1, kPre1, kPre1, dz2, kPre2, kPre2, G + PL.enc_pre2.w, kPre2, M1, M1, 0, s, 1, G + PL.enc_pre2.b));
    TACO_TRY(launch_conv_gemm(dense_problem(dz2, kPre2, PT + TL.enc_pre2, kPre1, nullptr, dz1, kPre1, M1, kPre1, kPre2,
                                            TACO_ACT_NONE), s));
    TACO_TRY(launch_act_bwd(ws + W.p1, dz1, enc_keep1, dz1, (int64_t)M1 * kPre1, TACO_ACT_RELU, s));
    TACO_TRY(tn(ws + W.emb, kEmbed, kEmbed, dz1, kPre1, kPre1, G + PL.enc_pre1.w, kPre1, M1, M1, 0, s, 1, G + PL.enc_pre1.b));
    TACO_TRY(launch_conv_gemm(dense_problem(dz1, kPre1, PT + TL.enc_pre1, kEmbed, nullptr, dEmb, kEmbed, M1, kEmbed, kPre1,
                                            TACO_ACT_NONE), s));
  }
  TACO_TRY(launch_embedding_bwd(dEmb, text, G + PL.emb, M1, shape->V, s));
  g_tn_side = nullptr;
  TACO_TRY(side_join(s, side));
  return record_segment(0, s);
}

extern "C" int taco_grad_segments(const TacoShape* shape, int64_t* bounds) {
  TACO_TRY(validate_shape(shape));
  TACO_REQUIRE(bounds != nullptr, "taco_grad_segments: null bounds");
  const ParamLayout& PL = layouts_for(*shape).P;
  bounds[0] = 0;
  bounds[1] = PL.enc.bank_w[0];    // [0, enc bank): embedding(s) + encoder pre_net (final LAST)
  bounds[2] = PL.enc.p1_w;         // [enc bank, enc proj1): encoder conv bank (its weight gradients: the last big launch of the pass)
  bounds[3] = PL.mem_w;            // [enc proj1, mem_w): encoder projections, highways, bi-GRU
  bounds[4] = PL.post.bank_w[0];   // [mem_w, post): attention memory layer + decoder
  bounds[5] = PL.total;            // [post, total): post-net CBHG + final dense (final FIRST)
  return kGradSegments;
}

extern "C" int taco_wait_grad_segment(int seg, void* stream) {
  TACO_REQUIRE(seg >= 0 && seg < kGradSegments, "taco_wait_grad_segment: segment %d out of range", seg);
  SideStream& x = side_stream();
  TACO_REQUIRE(x.seg_recorded && x.ev_seg_use[seg], "taco_wait_grad_segment: no taco_backward was issued by this thread on this device");
  if (hipStreamWaitEvent(as_stream(stream), x.ev_seg_use[seg], 0) != hipSuccess) {
    taco_set_error("taco_wait_grad_segment: hipStreamWaitEvent failed");
    return TACO_ELAUNCH;
  }
  return TACO_OK;
}

extern "C" int taco_conv_gemm(const float* A, int lda, const float* W, int ldw, const float* bias, const float* scale,
                              const float* shift, const float* residual, int ldr, const uint8_t* keep, float* C, int ldc,
                              float* Cpre, int M, int N, int K, int taps, int T, int pad_l, int act, void* stream) {
  ConvGemmProblem p;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.scale = scale; p.shift = shift; p.residual = residual;
  p.ldr = ldr; p.keep = keep; p.C = C; p.ldc = ldc; p.Cpre = Cpre; p.M = M; p.N = N; p.K = K; p.taps = taps; p.T = T;
  p.pad_l = pad_l; p.act = act;
  return launch_conv_gemm(p, as_stream(stream));
}

// debug / test aid: a dense layer whose weight rows are stored zero-padded to `nld` loadable columns (nld >= N, multiple of 4,
// <= ldw) -- how taco_forward runs the final 256 -> 1025 layer (padded weight copy, output pitch 1025)
extern "C" int taco_debug_conv_gemm_nld(const float* A, int lda, const float* W, int ldw, int nld, const float* bias, float* C,
                                        int ldc, int M, int N, int K, int act, void* stream) {
  TACO_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && nld >= N && nld <= ldw, "conv_gemm_nld: bad arguments");
  ConvGemmProblem p;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.Nld = nld; p.bias = bias; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.taps = 1; p.T = M; p.pad_l = 0; p.act = act;
  return launch_conv_gemm(p, as_stream(stream));
}

// debug / tuning aid: the tap- / k-split form of taco_conv_gemm used for the tall-skinny CBHG projections (slabs: scratch)
extern "C" int taco_debug_conv_gemm_ksplit(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                                           int M, int N, int K, int taps, int T, int pad_l, int act, float* slabs,
                                           int64_t slab_floats, void* stream) {
  TACO_REQUIRE(A && W && C && slabs && M > 0 && N > 0 && K > 0 && taps > 0 && T > 0, "conv_gemm_ksplit: bad arguments");
  ConvGemmProblem p;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.taps = taps;
  p.T = T; p.pad_l = pad_l; p.act = act;
  return launch_conv_gemm_tapsplit(p, slabs, slab_floats, as_stream(stream));
}

extern "C" int taco_gemm_tn(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, int M, int N, int K,
                            int taps, int T, int pad_l, int accumulate, void* stream) {
  GemmTnArgs a;
  a.A = A; a.lda = lda; a.Y = dY; a.ldy = ldy; a.W = dW; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.taps = taps; a.T = T;
  a.pad_l = pad_l;
  return launch_gemm_tn(a, accumulate == 0, as_stream(stream));
}

extern "C" int taco_debug_gemm_naive(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                                     int M, int N, int K, int taps, int T, int pad_l, int act, void* stream) {
  TACO_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && taps > 0 && T > 0, "gemm_naive: bad arguments");
  ConvGemmProblem p;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.taps = taps;
  p.T = T; p.pad_l = pad_l; p.act = act;
  return launch_gemm_naive(p, as_stream(stream));
}

extern "C" int taco_bigru_fwd(const float* x, const float* wg_fw, const float* bg_fw, const float* wc_fw, const float* bc_fw,
                              const float* wg_bw, const float* bg_bw, const float* wc_bw, const float* bc_bw, float* xg,
                              float* out, float* ruc, int B, int T, void* stream) {
  TACO_REQUIRE(x && wg_fw && bg_fw && wc_fw && bc_fw && wg_bw && bg_bw && wc_bw && bc_bw && xg && out,
               "bigru_fwd: null pointer argument");
  hipStream_t s = as_stream(stream);
  const int M = B * T;
  BiGruWeights w;
  w.wg[0] = wg_fw; w.bg[0] = bg_fw; w.wc[0] = wc_fw; w.bc[0] = bc_fw;
  w.wg[1] = wg_bw; w.bg[1] = bg_bw; w.wc[1] = wc_bw; w.bc[1] = bc_bw;
  ConvGemmBatch batch;
  batch.n = 4;
  for (int d = 0; d < 2; ++d) {
    batch.p[2 * d] = dense_problem(x, kCb, w.wg[d], 2 * kCb, w.bg[d], xg + d * 3 * kCb, 6 * kCb, M, 2 * kCb, kCb, TACO_ACT_NONE);
    batch.p[2 * d + 1] = dense_problem(x, kCb, w.wc[d], kCb, w.bc[d], xg + d * 3 * kCb + 2 * kCb, 6 * kCb, M, kCb, kCb, TACO_ACT_NONE);
  }
  TACO_TRY(launch_conv_gemm_batch(batch, s));
  return launch_bigru_fwd(xg, w, nullptr, out, ruc, B, T, s);
}

extern "C" int taco_clip_adam_step_guarded(float* params, const float* grads, float* m, float* v, int64_t n, float lr,
                                           float cap, int64_t step, float* scratch, float* gnorm_out,
                                           const int32_t* err_words, void* stream) {
  TACO_REQUIRE(params && grads && m && v && scratch && n > 0 && step >= 1, "clip_adam_step: bad arguments");
  hipStream_t s = as_stream(stream);
  TACO_TRY(launch_sumsq(grads, n, scratch, s));
  return launch_clip_adam(params, grads, m, v, n, lr, cap, step, scratch, gnorm_out, err_words, s);
}

extern "C" int taco_clip_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float cap,
                                   int64_t step, float* scratch, float* gnorm_out, void* stream) {
  return taco_clip_adam_step_guarded(params, grads, m, v, n, lr, cap, step, scratch, gnorm_out, nullptr, stream);
}

extern "C" int taco_clear_error(const TacoShape* shape, int train, void* workspace, void* stream) {
  TACO_TRY(validate_shape(shape));
  TACO_REQUIRE(workspace != nullptr, "taco_clear_error: null workspace");
  const Layouts& L = layouts_for(*shape);
  const WsLayout& W = train ? L.Wtrain : L.Winfer;
  hipError_t e = hipMemsetAsync(static_cast<float*>(workspace) + W.err, 0, 512 * sizeof(float), as_stream(stream));
  if (e != hipSuccess) {
    taco_set_error("taco_clear_error: memset: %s", hipGetErrorString(e));
    return TACO_ELAUNCH;
  }
  return TACO_OK;
}

extern "C" int taco_denorm_unframe(const float* output, const float* stft_mean, const float* stft_std, float* spec,
                                   float* mag_t, int B, int Td, int r, int C, void* stream) {
  TACO_REQUIRE(output && stft_mean && stft_std && (spec || mag_t) && B > 0 && Td > 0 && r >= 1 && r <= 5 && C > 0,
               "denorm_unframe: bad arguments");
  return launch_denorm_unframe(output, stft_mean, stft_std, spec, mag_t, B, Td, r, C, as_stream(stream));
}

extern "C" int64_t taco_griffinlim_workspace_bytes(int B, int F) {
  if (B <= 0 || F < 5) return TACO_EINVAL;   // launch_griffinlim needs F >= 5 (centre padding of 1024 samples at hop 300)
  return griffinlim_workspace_floats(B, F) * (int64_t)sizeof(float);
}

extern "C" int taco_griffinlim(const float* mag_t, const float* phase0, float* wave, void* workspace, int B, int F, int n_iter,
                               void* stream) {
  return launch_griffinlim(mag_t, phase0, wave, static_cast<float*>(workspace), B, F, n_iter, as_stream(stream));
}

extern "C" int taco_fill_bernoulli(uint8_t* out, int64_t n, float p_one, uint64_t seed, void* stream) {
  TACO_REQUIRE(out && n > 0, "fill_bernoulli: bad arguments");
  return launch_bernoulli(out, n, p_one, seed, as_stream(stream));
}

extern "C" int taco_debug_last_cluster(int which) { return decoder_last_cluster(which); }

extern "C" int taco_debug_spin(int blocks, int threads, int lds_bytes, int usec, void* stream) {
  return launch_spin(blocks, threads, lds_bytes, usec, as_stream(stream));
}

extern "C" int taco_debug_clock_probe(long long* out3, int iters, void* stream) {
  return launch_clock_probe(out3, iters, as_stream(stream));
}

extern "C" int taco_debug_fabric_probe(long long* out32, void* gran4k, const void* scratch, long long scratch_bytes, int iters, void* stream) {
  return launch_fabric_probe(out32, gran4k, scratch, scratch_bytes, iters, as_stream(stream));
}

extern "C" int taco_profile_enable(int mask) {
  g_prof_mask = mask & 31;
  return TACO_OK;
}

extern "C" int taco_profile_read2(int which, float* ms, double* flops, int cap) {
  TACO_REQUIRE(which >= 0 && which < 4, "profile_read: category %d out of range", which);
  ProfRing& r = g_prof[which];
  int n = 0;
  for (int i = 0; i < r.n; ++i) {
    if (hipEventSynchronize(r.stop[i]) != hipSuccess) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.start[i], r.stop[i]) != hipSuccess) break;
    if (ms && n < cap) ms[n] = t;
    if (flops && n < cap) flops[n] = r.flops[i];
    ++n;
  }
  r.n = 0;
  return n;
}

extern "C" int taco_debug_profile_labels(int which, char* buf, int cap) {
  TACO_REQUIRE(which >= 0 && which < 4 && buf && cap > 0, "profile_labels: bad arguments");
  ProfRing& r = g_prof[which];
  int pos = 0;
  for (int i = 0; i < r.n; ++i) {
    const int w = snprintf(buf + pos, cap - pos, "%s\n", r.label[i]);
    if (w < 0 || pos + w >= cap) break;
    pos += w;
  }
  buf[pos < cap ? pos : cap - 1] = 0;
  return r.n;
}

extern "C" int taco_profile_read(int which, float* ms, int cap) { return taco_profile_read2(which, ms, nullptr, cap); }
