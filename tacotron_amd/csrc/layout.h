// layout.h -- parameter / transposed-parameter / workspace layouts of libtaco_hip.so.
// The parameter order is the reference graph's variable order (tacotron.py:107-154) with TF variable layouts; it
// must match oracle/taco_numpy.py::param_spec (asserted by tests/test_layout.py).
#pragma once
#include <string>
#include <vector>

#include "common.h"

struct DenseP {
  int64_t w = -1, b = -1;
  int in = 0, out = 0;
};
struct GruP {
  int64_t wg = -1, bg = -1, wc = -1, bc = -1;
  int cin = 0, h = 0;
};
struct CbhgP {
  int K = 0, cin = 0, c1 = 0, c2 = 0;
  int64_t bank_w[16], bank_b[16];
  int64_t bank_g = -1, bank_be = -1;
  int64_t p1_w = -1, p1_b = -1, p1_g = -1, p1_be = -1;
  int64_t p2_w = -1, p2_b = -1, p2_g = -1, p2_be = -1;
  bool spk = false;            // encoder CBHG of a multi-speaker model: speaker sites of ops.py:101-115
  bool has_adapt[4] = {false, false, false, false};
  DenseP adapt[4];             // highway() input adapter: post-net layer 0 (80->128); every layer (256->128) with speakers
  DenseP spkd[4];              // per-layer speaker dense (16->128)
  DenseP gru_init;             // speaker dense (16->128) -> bi-GRU initial state
  DenseP hwT[4], hwH[4];
  GruP fw, bw;
};
struct ParamLayout {
  int64_t emb = -1;
  int64_t spk_embed = -1;  // (S,16) when S > 1
  DenseP enc_pre1, enc_pre2;
  CbhgP enc;
  int64_t mem_w = -1;
  DenseP dec_pre1, dec_pre2, in_proj;
  GruP gru[3];
  DenseP out_proj;
  int64_t q_w = -1, att_v = -1, att_w = -1;
  CbhgP post;
  DenseP post_dense;
  int64_t total = 0;
  std::vector<TacoTensorInfo> rows;
};

// Transposed (and tap-flipped) copies used by the backward pass; offsets into the workspace "paramsT" region.
struct CbhgT {
  int64_t bank[16];      // (k, 128, cin)
  int64_t p1 = -1;       // (3, c1, K*128)
  int64_t p2 = -1;       // (3, c2, c1)
  int64_t adapt[4];      // (128, cin_h): transposed rows [0, cin_h) of the adapter (cin_h = c2, or 128 with speakers)
  int64_t adapt_s[4];    // (128, 128): transposed rows [128, 256) of the adapter (speaker half)
  int64_t spkd[4];       // (128, 16)
  int64_t gru_init = -1; // (128, 16)
  int64_t hw[4];         // (256, 128): rows [0,128) = Wt^T, [128,256) = Wh^T
  int64_t gru_x = -1;    // (768, 128): [Wg_fw[:128]^T ; Wc_fw[:128]^T ; Wg_bw[:128]^T ; Wc_bw[:128]^T]
  int64_t wghT[2];       // (256, 128) = Wg[128:]^T
  int64_t wchT[2];       // (128, 128) = Wc[128:]^T
};
struct TransLayout {
  int64_t enc_pre1 = -1, enc_pre2 = -1;
  CbhgT enc;
  int64_t mem_w = -1;
  int64_t dec_pre1 = -1, dec_pre2 = -1, in_proj = -1;
  int64_t gw[3], cw[3];
  int64_t out_proj = -1, q_w = -1, att_w = -1;
  CbhgT post;
  int64_t post_dense = -1;
  int64_t total = 0;
};

struct CbhgWs {
  int64_t bank, pool, pj1pre, pj1, pj2pre, res, hx[4], h[5], th[4], xg, out, ruc;
  int64_t sv[4], rowb[4], h0;  // speaker sites: relu(dense(spk)) (B,128), per-sequence adapter bias (B,128), GRU init (B,128);
                               // sv[0..3] and h0 are ONE contiguous (5,B,128) block (one ReLU-backward launch over all of it)
  int64_t dh0, dsmall, dsmall2; // backward: (2,B,128), (B,128), (B,128)
  int64_t dsm[4], dsm2[5], dspk_part;   // backward, round 6: per-layer d rowb (B,128); d[sv | h0] (5,B,128) contiguous; (5,B,16) partial d spk_e
};
struct WsLayout {
  // forward
  int64_t emb, p1, p2, spk_e, dspk_e;
  CbhgWs enc;
  int64_t values, keys, vwx, vwg, vcen, vwxc;
  int64_t stash, prein, xchg, err;
  // decoder composite weights (products of consecutive linear maps; rebuilt every call, see build_dec_composites)
  int64_t dc_wx, dc_wg0, dc_bg0, dc_wo, dc_bo, dc_wp1o, dc_bp1o;
  CbhgWs post;
  int64_t wd_pad;
  int64_t tapsplit;   // 4 x max(M1*128, M2*256) floats: per-chunk partial sums of the proj1 convolutions (conv_gemm_tapsplit)
  int64_t tapsplit_floats = 0;
  int64_t loss;  // 4 floats
  // backward
  int64_t ds2s, dout_pad, paramsT, gstash, dkeys, dvalues, ds2s_tot;
  int64_t bc_wxct, bc_wdx, bc_wmx;
  int64_t bc_fa, bc_wot, bc_g, bc_h1, bc_h2, bc_cq, bc_cp;   // decoder backward composites and small weight-gradient factors
  int64_t dattv;      // (B,256) per-row attention_v gradient partials
  int64_t post_dpj1, post_dz1, post_dpool, post_dx;
  int64_t enc_dpj1, enc_dz1, enc_dpool, enc_dx, pre_dz2, pre_dz1, pre_demb;   // likewise for the encoder CBHG / pre-net (side-stream weight gradients)   // post-net backward operands that outlive cbhg_bwd (deferred weight gradients)
  int64_t gA, gB, gC, gD, gE, gF, gG, scratch;
  int64_t wimg = -1;  // pre-split weight images (for_each_weight_image order; each 16-byte aligned)
  int64_t total = 0;  // floats
  std::vector<TacoTensorInfo> rows;
};

// Which weights get a pre-split bf16 plane image (kernels.h "weight images"): the B operands of the NN launches that run on
// gemm2.hip's bf16x3 form.  f(key_src, key_off, key_ldw, data_src, data_off, data_ldw, taps, K, N, backward):
//   *_src: 0 = parameter buffer, 1 = workspace paramsT region, 2 = workspace wd_pad; key = the pointer / pitch the GEMM launch
//   passes (what the table is searched by), data = where the builder reads the values (post/dense: the parameters themselves --
//   wd_pad is a re-pitched copy made by a side-stream launch that may not have run yet).  One list for the workspace layout
//   (sizes) and for model.hip (registration + build), so the two cannot drift apart.
template <class F>
inline void for_each_weight_image(const ParamLayout& P, const TransLayout& T, bool train, F f) {
  const CbhgP* cp[2] = {&P.enc, &P.post};
  const CbhgT* ct[2] = {&T.enc, &T.post};
  for (int i = 0; i < 2; ++i) {
    const CbhgP& c = *cp[i];
    for (int k = 1; k <= c.K; ++k) f(0, c.bank_w[k - 1], kCb, 0, c.bank_w[k - 1], kCb, k, c.cin, kCb, false);   // conv bank, width k
    f(0, c.p1_w, c.c1, 0, c.p1_w, c.c1, 3, c.K * kCb, c.c1, false);                                           // proj1 (k-split)
    const GruP* g[2] = {&c.fw, &c.bw};
    for (int d = 0; d < 2; ++d) {                                                                           // bi-GRU x-projections
      f(0, g[d]->wg, 2 * kCb, 0, g[d]->wg, 2 * kCb, 1, kCb, 2 * kCb, false);
      f(0, g[d]->wc, kCb, 0, g[d]->wc, kCb, 1, kCb, kCb, false);
    }
  }
  f(2, 0, 1028, 0, P.post_dense.w, kFft, 1, 2 * kCb, kFft, false);                                            // final dense layer
  if (!train) return;
  for (int i = 0; i < 2; ++i) {
    const CbhgP& c = *cp[i];
    const CbhgT& t = *ct[i];
    f(1, t.bank[0], c.cin, 1, t.bank[0], c.cin, c.K * (c.K + 1) / 2, kCb, c.cin, true);     // bank input gradient: all widths' taps, one reduction
    f(1, t.p1, c.K * kCb, 1, t.p1, c.K * kCb, 3, c.c1, c.K * kCb, true);                    // proj1 input gradient (d pool)
    f(1, t.gru_x, kCb, 1, t.gru_x, kCb, 1, 6 * kCb, kCb, true);                             // bi-GRU x-projection input gradient
  }
  f(1, T.post_dense, 2 * kCb, 1, T.post_dense, 2 * kCb, 1, 1028, 2 * kCb, true);            // final dense input gradient (K padded to 1028)
}
inline int64_t weight_image_floats(int taps, int K, int N) {   // = kernels.h weight_image_bytes / 4 (12 KB per (n-tile, 16-deep k-tile))
  return (int64_t)((N + 127) / 128) * taps * (((K + 31) / 32) * 2) * 3072;
}

void build_param_layout(const TacoShape& s, ParamLayout& L);
void build_trans_layout(const TacoShape& s, const ParamLayout& P, TransLayout& T);
void build_ws_layout(const TacoShape& s, bool train, const TransLayout& T, WsLayout& W);
int validate_shape(const TacoShape* s);
