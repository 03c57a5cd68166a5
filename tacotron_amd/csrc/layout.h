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
  int64_t sv[4], rowb[4], h0;  // speaker sites: relu(dense(spk)) (B,128), per-sequence adapter bias (B,128), GRU init (B,128)
  int64_t dh0, dsmall, dsmall2; // backward: (2,B,128), (B,128), (B,128)
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
  int64_t total = 0;  // floats
  std::vector<TacoTensorInfo> rows;
};

void build_param_layout(const TacoShape& s, ParamLayout& L);
void build_trans_layout(const TacoShape& s, const ParamLayout& P, TransLayout& T);
void build_ws_layout(const TacoShape& s, bool train, const TransLayout& T, WsLayout& W);
int validate_shape(const TacoShape* s);
