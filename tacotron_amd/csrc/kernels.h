// kernels.h -- internal launch interface shared by the .hip translation units of libtaco_hip.so.
#pragma once
#include "common.h"

// ---------------------------------------------------------------- profiling rings (model.hip)
// category 0 / 1: decoder fwd / bwd kernel, 2: MFMA GEMM family, 3: bi-GRU recurrences.  begin returns a slot (or -1 when the
// category is not being recorded); end stamps the stop event and the launch's algorithmic FLOPs.
int taco_prof_begin(int which, hipStream_t s);
void taco_prof_end(int which, int slot, hipStream_t s, double flops);
void taco_prof_label(int which, int slot, const char* fmt, ...) __attribute__((format(printf, 3, 4)));   // no-op when slot < 0

// ---------------------------------------------------------------- gemm.hip
constexpr int kMaxGemmBatch = 16;

struct ConvGemmProblem {
  const float* A = nullptr;
  const float* W = nullptr;
  const float* bias = nullptr;
  const float* scale = nullptr;
  const float* shift = nullptr;
  const float* residual = nullptr;
  const uint8_t* keep = nullptr;
  float* C = nullptr;
  float* Cpre = nullptr;
  int lda = 0, ldw = 0, ldr = 0, ldc = 0;
  int M = 0, N = 0, K = 0, taps = 1, T = 1, pad_l = 0, act = 0, flags = 0;
  int bias_stride = 0; // 0: one bias vector; else bias[(m / T) * bias_stride + n] (per-sequence bias, e.g. speaker sites)
  int Nld = 0;         // loadable W columns (>= N, multiple of 4, <= ldw) when the storage is padded; 0 = derive from N
  int atomic_out = 0;  // C += result with fp32 atomics (C pre-zeroed by the caller); several problems may share C
  float scale_mul = 1.f;   // scale[n] is multiplied by this (BN inference: scale = gamma, scale_mul = 1/sqrt(1+eps))
  int it0 = 0, it1 = 0;    // gemm2.hip only: restrict the (tap, 32-deep k-tile) sequence to [it0, it1) (k-split chunk); it1 = 0: all
  // gemm2.hip only: the conv BANK's input gradient as ONE reduction (ops.py:54-62 backwards): `taps` = F (F + 1) / 2 counts the taps
  // of all F = bank_filters 'same' convolutions of width f = 1..F in order; global tap g = f (f - 1) / 2 + j reads the rows shifted
  // by j - ((f - 1) - (f - 1) / 2) from the 128-column block f - 1 of A (A = d bank, lda = F * 128) and the weights W + g * K * ldw
  // (the transposed kernels of all widths, contiguous).  pad_l must be the largest backward pad, (F - 1) - (F - 1) / 2.
  int bank_filters = 0;
  // gemm2.hip only: max_pool1d(2, stride 1, 'same') along the sequence in the epilogue (ops.py:64-71).  C[m] receives
  // max(z[m], z[m + 1]) (z[m] alone on the last row of a sequence) of the affine'd activation z; Cpre, if given, the
  // activation before the affine.  m-tiles overlap by one row (stride 127) so every pooled row has its successor in the tile.
  int pool = 0;
  // pool == 2: the BACKWARD of that op in the epilogue of the GEMM that produces d(pooled) (the proj1 input gradient): the
  // accumulator row m is dy[m]; with z = x scale + shift (x = pool_x, the stored activations, row pitch ldc),
  //   dz[m] = [last row of its sequence or z[m] >= z[m+1]] dy[m] + [not first and z[m] > z[m-1]] dy[m-1],
  //   C[m] = (x[m] > 0) dz[m] scale   (through the ReLU that produced x),  pool_dgamma += sum_m dz x / sqrt(1+eps),  pool_dbeta += sum_m dz.
  // Tiles overlap by one row at the TOP (stride 127): a tile owns its rows 1..127 (tile 0 also row 0).
  const float* pool_x = nullptr;
  float* pool_dgamma = nullptr;
  float* pool_dbeta = nullptr;
  // gemm2.hip only, filled by launch_conv_gemm2 from the weight-image table (weight_image_find): the pre-split bf16 plane image
  // of W and its k-tiles (of 16) per n-tile; null = split W in registers
  const void* Wimg = nullptr;
  int img_its = 0;
};
struct ConvGemmBatch {
  ConvGemmProblem p[kMaxGemmBatch];
  int n = 0;
};
struct GemmTnArgs {
  const float* A = nullptr;
  const float* Y = nullptr;
  float* W = nullptr;
  float* dbias = nullptr;  // optional: dbias[n] += sum_m Y[m][n] (bias gradient), accumulated by the k-tile-0 / tap-0 blocks
  int lda = 0, ldy = 0, ldw = 0;
  int M = 0, N = 0, K = 0, taps = 1, T = 1, pad_l = 0;
  int batch = 1;
  int64_t strideA = 0, strideY = 0, strideW = 0;
  int splits = 1, chunk = 0, flags = 0;
  int Nld = 0;   // loadable Y columns (>= N, multiple of 4, <= ldy) when Y rows are zero-padded; 0 = derive from N
  int xcd = 0;   // 1: XCD-aware block order (set by the launcher, gemm.hip `tn_xcd_order`)
  int ktap = 0;  // > 0 (set by the launcher): the taps are MERGED into the K dimension -- K holds taps * ktap, column kk of a tile is
                 // column kk % ktap of tap kk / ktap (its rows shifted accordingly), taps == 1.  For K that is no multiple of the 64-row
                 // tile (post-net conv bank: K = 80 -> 2 tiles of 64 per tap, 37.5 % of them padding; merged: 640 rows = 10 tiles for 8 taps)
};

int launch_conv_gemm_batch(ConvGemmBatch& batch, hipStream_t stream);
// gemm2.hip: DMA-staged NN kernel for batches that meet the vector contract (flags == 3 on every problem) and have at least
// TACO_GEMM2_MIN_TILES (default 96) 128 x 128 tiles; returns TACO_ENOTFOUND without launching otherwise.
int launch_conv_gemm_slab_sum(const ConvGemmProblem& p, const float* slabs, int n, hipStream_t stream);   // C = epilogue(sum of n (M,N) slabs)
void conv_gemm_set_flags(ConvGemmProblem& p);   // derives the vector-contract bits (flags & 3) and Nld
int launch_conv_gemm2(ConvGemmBatch& batch, hipStream_t stream, bool force = false);
// true when launch_conv_gemm2 would take this batch (flags already set): callers that want the pooled epilogue ask first and
// run conv + bn_maxpool as two launches otherwise
bool conv_gemm2_would_launch(const ConvGemmBatch& batch);
int gemm2_min_tiles();   // TACO_GEMM2_MIN_TILES (0 disables gemm2.hip)
// ---- pre-split weight images (gemm2.hip, round 6).  The bf16x3 form of the NN kernel needs the three bf16 planes (h, m, l) of both
// operands; splitting the B (weight) tile costs every wave 176 VALU operations per 24 MFMAs, identically in all four waves of a
// workgroup and in every workgroup that reads the tile.  The weights change once per step, so their planes are formed ONCE per
// step into an IMAGE the kernel can DMA and read as ready MFMA fragments:
//   image(W)[n-tile][k-tile it][plane P = h, m, l][sub-tile j][kh][li][q]  (bf16),  k-tile = 16 k-values of one tap, it = tap * kt + k0 / 16,
//   element = plane P of W[tap][16 kt' + 8 (q >> 2) + 4 kh + (q & 3)][128 n-tile + 4 li + j], zero outside K x N --
//   12 KB per (n-tile, k-tile), contiguous in `it` (taps, k-split chunks and the conv bank's global tap index are all runs of it).
// The table is per host thread and is rebuilt by every model-level entry point (taco_forward / taco_backward / taco_infer) from
// the parameter and workspace pointers of THAT call; the images themselves live in the caller's workspace, are built by
// taco_forward / taco_infer and stay valid until the parameters change (like the transposed copies in `paramsT`).
constexpr int kWImgTileBytes = 3 * 16 * 128 * 2;
int64_t weight_image_bytes(int taps, int K, int N);                          // size of image(W) for W of (taps, K, N)
void weight_images_clear();                                                  // empties the table (and the queue of unbuilt jobs)
// registers image(W) at `img` (16-byte aligned, weight_image_bytes large) and queues its build; N = loadable columns of W (<= ldw)
// src / src_ldw (optional): where the builder reads the values when that is not W itself (a re-pitched copy that is made later)
int weight_image_add(const float* W, int ldw, int taps, int K, int N, void* img, bool queue_build, const float* src = nullptr, int src_ldw = 0);
int weight_images_build(hipStream_t s);                                      // one launch per <= 40 queued jobs
const void* weight_image_find(const float* W, int ldw, int taps, int K, int N, int* its);   // null when W has no (matching) image
// The CBHG's highway layers as one launch (highway.hip).  Layer l: th[l] = [sigmoid(x Wt+bt) | relu(x Wh+bh)] (M,256),
// y[l] = H*T + x*(1-T) (M,128) feeds layer l+1.
struct HighwayStackArgs {
  const float* x = nullptr;    // (M,128) input of the first layer
  const float* wt[4]; const float* bt[4]; const float* wh[4]; const float* bh[4];
  float* th[4];               // (M,256) [T | H] stash per layer, or null (inference: no backward pass will read it)
  float* y[4];
  int M = 0, nl = 0;
  // adapter form (multi-speaker encoder; all four layers or none): x_l = h_l . wa[l] + rowb[l][m / T], stashed in hx[l], feeds the gates
  const float* wa[4] = {nullptr, nullptr, nullptr, nullptr};     // (128,128): rows [0,128) of the layer's adapter kernel
  const float* rowb[4] = {nullptr, nullptr, nullptr, nullptr};   // (B,128) per-sequence bias
  float* hx[4] = {nullptr, nullptr, nullptr, nullptr};           // (M,128)
  int T = 0;                                                     // rows per sequence
};
int launch_highway_stack_fwd(const HighwayStackArgs& a, hipStream_t s);
// prenet.hip: the encoder pre_net (two dense + ReLU + dropout layers) and its activation-gradient chain as one launch each.
//   forward : x (M,256) -> y1 = drop1(relu(x w1 + b1)) (M,256) -> y2 = drop2(relu(y1 w2 + b2)) (M,128)
//   backward: x = d p2 (M,128), y2_in = p2, y1_in = p1, w1 = W2^T (128,256), w2 = W1^T (256,256):
//             x_out = dz2 (M,128), y1 = dz1 (M,256), y2 = d input (M,256); keep1 / keep2 are the masks of layer 1 / layer 2
struct PrenetArgs {
  const float* x = nullptr;
  const float* w1 = nullptr; const float* b1 = nullptr;
  const float* w2 = nullptr; const float* b2 = nullptr;
  const uint8_t* keep1 = nullptr; const uint8_t* keep2 = nullptr;   // (M, width of layer 1 / layer 2) or null: no dropout
  const float* y1_in = nullptr; const float* y2_in = nullptr;       // backward only: the forward activations (ReLU masks)
  float* x_out = nullptr;                                            // backward only
  float* y1 = nullptr; float* y2 = nullptr;
  int M = 0;
  long long* trace = nullptr;   // probe builds (-DTACO_PN_TRACE) only
};
int launch_prenet_fwd(const PrenetArgs& a, hipStream_t s);
int launch_prenet_bwd(const PrenetArgs& a, hipStream_t s);
struct HighwayStackBwdArgs {
  const float* g = nullptr;    // (M,128) dL/d output of the last layer
  const float* wT[4];          // (256,128) [Wt^T ; Wh^T] per layer
  const float* th[4];          // forward stash [T | H]
  const float* x[4];           // layer inputs
  float* dth[4];               // (M,256) d[T|H] pre-activation gradients (operands of the weight-gradient GEMMs)
  float* gout = nullptr;       // (M,128) dL/d input of layer 0
  int M = 0, nl = 0;
  // adapter form: x[l] is the adapter OUTPUT hx[l]; dhx[l] receives dL/d hx[l]; the chain continues through waT[l] = wa[l]^T
  const float* waT[4] = {nullptr, nullptr, nullptr, nullptr};    // (128,128)
  float* dhx[4] = {nullptr, nullptr, nullptr, nullptr};          // (M,128)
};
int launch_highway_stack_bwd(const HighwayStackBwdArgs& a, hipStream_t s);
int launch_conv_gemm(const ConvGemmProblem& p, hipStream_t stream);
// A taps-wide convolution whose tile grid cannot fill the chip (tall-skinny: N <= 256) run as `taps` independent one-tap
// problems into `slabs` (taps x M x N floats) plus ONE elementwise pass that adds the slabs in tap order and applies the
// epilogue: taps x the workgroups, results independent of scheduling (no atomics).  Falls back to launch_conv_gemm when the
// split does not pay.
int launch_conv_gemm_tapsplit(const ConvGemmProblem& p, float* slabs, int64_t slab_floats, hipStream_t stream);
int launch_gemm_tn(GemmTnArgs a, bool zero_first, hipStream_t stream);
constexpr int kMaxTnBatch = 26;
struct GemmTnBatch {
  GemmTnArgs p[kMaxTnBatch];
  int first[kMaxTnBatch];   // first linear block of each problem (filled by the launcher)
  int gx[kMaxTnBatch], gy[kMaxTnBatch];
  int n = 0;
};
// Launches every queued problem (accumulating into W) in as few grids as possible and empties the batch.
int launch_gemm_tn_batch(GemmTnBatch& b, hipStream_t stream);
// gemm2.hip: second-generation weight-gradient kernel (accumulating; 128 x 128 tiles, DMA-staged 32-row stages)
bool gemm_tn2_eligible(const GemmTnArgs& a);
int launch_gemm_tn2(const GemmTnArgs* probs, int n, hipStream_t stream);
int launch_gemm_naive(const ConvGemmProblem& p, hipStream_t stream);

// ---------------------------------------------------------------- elementwise.hip
int launch_embedding(const float* table, const int32_t* ids, float* out, int64_t rows, int V, hipStream_t s, int width = kEmbed);
int launch_mask_pos(float* g, int ldg, const float* y, int ldy, int64_t rows, int cols, float scale, hipStream_t s);
int launch_center_rows(const float* x, const int32_t* len, float* out, int B, int T, int C, hipStream_t s);
int launch_embedding_bwd(const float* dout, const int32_t* ids, float* dtable, int64_t rows, int V, hipStream_t s, int width = kEmbed);
// y = maxpool2_same(x*scale+shift) along T; x,y (B*T, C)
int launch_bn_maxpool(const float* x, const float* gamma, const float* beta, float* y, int B, int T, int C, hipStream_t s);
// dgamma, dbeta of the op above (atomics) and dx THROUGH the ReLU that produced x: dx = (x > 0) * d(x*scale+shift) * scale
int launch_bn_maxpool_bwd(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                          float* dgamma, float* dbeta, int B, int T, int C, hipStream_t s);
// highway combine: th (M,256) = [T | H]; y = H*T + x*(1-T)
int launch_highway_combine(const float* th, const float* x, float* y, int64_t M, hipStream_t s);
// dth (M,256) = [dy*(H-x)*T(1-T) | dy*T*(H>0)], dx = dy*(1-T)
int launch_highway_combine_bwd(const float* th, const float* x, const float* dy, float* dth, float* dx, int64_t M,
                               hipStream_t s);
// dz = dy * act'(y) [* keep*2]; in place allowed (dz == dy)
int launch_act_bwd(const float* y, const float* dy, const uint8_t* keep, float* dz, int64_t n, int act, hipStream_t s);
// BN-affine backward around a stored pre-affine activation:
//   dgamma += sum_m dy*pre*rs, dbeta += sum_m dy, dz = dy*gamma*rs*act'(pre)   (rs = 1/sqrt(1+eps); act in {none, relu})
int launch_affine_act_bwd(const float* pre, const float* gamma, const float* dy, float* dz, float* dgamma, float* dbeta,
                          int64_t M, int N, int act, hipStream_t s);
// out[b*N + n] = sum_{t<T} x[(b*T + t)*ld + n]   (overwrites)
int launch_colsum_batched(const float* x, int ld, float* out, int B, int T, int N, hipStream_t s);
// values = enc * (t < len[b])
int launch_mask_rows(const float* x, const int32_t* len, float* y, int B, int T, int C, hipStream_t s);
int launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s);  // y = a + b
// L1 losses + sign gradients.  loss_parts[0..kLossParts) = per-block partial sums of |a-b| (overwritten; summed in block
// order by launch_finish_loss: no atomics, reproducible).  grad (ldg >= N) = sign(a-b), pad columns zeroed.
constexpr int kLossParts = 512;
int launch_l1(const float* a, const float* b, float* grad, int ldg, float* loss_parts, int64_t M, int N, hipStream_t s);
// parts (2 x kLossParts): loss[1] = sum parts[0..), loss[2] = sum parts[kLossParts..), loss[0] = loss[1] + loss[2]; out (nullable) = copy
int launch_finish_loss(float* loss, const float* parts, float* out, hipStream_t s);
// Batched form: up to kMaxTransposeBatch (in, out, taps, K, N) jobs in ONE launch.
constexpr int kMaxTransposeBatch = 96;
struct TransposeJob {
  const float* in;
  float* out;
  int taps, K, N, tile0;   // tile0 = first linear tile index of this job (filled by the launcher)
  int ldi = 0, ldo = 0;    // input / output pitch (0: dense, N / K); sub-blocks of larger matrices when set (taps must be 1)
};
struct TransposeBatch {
  TransposeJob j[kMaxTransposeBatch];
  int n = 0;
};
int launch_transpose_batch(TransposeBatch& b, hipStream_t s);
// inverse r-frame layout + de-normalisation (+ optional exp / transpose for the vocoder); see the kernel
int launch_denorm_unframe(const float* out, const float* mean, const float* stdv, float* spec, float* mag_t, int B, int Td,
                          int r, int C, hipStream_t s);
constexpr int kSumsqParts = 256;
int launch_sumsq(const float* x, int64_t n, float* out, hipStream_t s);  // out[0..kSumsqParts) = per-block sums of x^2 (overwritten)
// err (nullable): two int32 decoder error words; if either is non-zero the update is skipped (gnorm_out = -1)
int launch_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float cap, int64_t step,
                     const float* sumsq, float* gnorm_out, const int32_t* err, hipStream_t s);
int launch_bernoulli(uint8_t* out, int64_t n, float p_one, uint64_t seed, hipStream_t s);
int launch_spin(int blocks, int threads, int lds_bytes, int usec, hipStream_t s);
int launch_clock_probe(long long* out, int iters, hipStream_t s);   // out[0] = shader cycles, out[1] = 100 MHz ticks of `iters` dependent FMAs
int launch_fabric_probe(long long* out32, void* gran4k, const void* scratch, int64_t scratch_bytes, int iters, hipStream_t s);
// Batched buffer initialisation: up to kMaxInitJobs zero-fills / (strided) copies of fp32 blocks in ONE launch instead of one
// runtime memset / memcpy launch each (~5 us apiece, serialised on their stream).  Jobs of a batch must not overlap.
constexpr int kMaxInitJobs = 16;
struct InitJob {
  float* dst = nullptr;
  const float* src = nullptr;   // null: fill with zeros
  int rows = 1;
  int64_t cols = 0;             // floats per row
  int64_t ldd = 0, lds = 0;     // row pitches (floats)
  int blk0 = 0;                 // first block of the job (filled by the launcher)
};
struct InitBatch {
  InitJob j[kMaxInitJobs];
  int n = 0;
  int fill(float* dst, int64_t floats) { return fill2d(dst, 1, floats, floats); }
  int fill2d(float* dst, int rows, int64_t cols, int64_t ldd) { return add(dst, nullptr, rows, cols, ldd, 0); }
  int copy(float* dst, const float* src, int64_t floats) { return add(dst, src, 1, floats, floats, floats); }
  int copy2d(float* dst, int64_t ldd, const float* src, int64_t lds, int rows, int64_t cols) { return add(dst, src, rows, cols, ldd, lds); }
  int add(float* dst, const float* src, int rows, int64_t cols, int64_t ldd, int64_t lds) {
    if (n >= kMaxInitJobs) return TACO_EINVAL;
    if (rows <= 0 || cols <= 0) return TACO_OK;
    InitJob& q = j[n++];
    q.dst = dst; q.src = src; q.rows = rows; q.cols = cols; q.ldd = ldd; q.lds = lds;
    return TACO_OK;
  }
};
int launch_init_batch(InitBatch& b, hipStream_t s);   // communication-kernel stand-in (tests)

// ---------------------------------------------------------------- bigru.hip
struct BiGruWeights {
  const float* wg[2];  // (256,256) rows [0,128) x-part, [128,256) h-part
  const float* bg[2];  // (256)
  const float* wc[2];  // (256,128)
  const float* bc[2];  // (128)
};
// xg (B,T,768): per direction d: [d*384, d*384+256) gates x-proj (+bias), [d*384+256, d*384+384) candidate x-proj (+bias)
// h0 (B,128): initial state of both directions (nullable = zeros)
int launch_bigru_fwd(const float* xg, const BiGruWeights& w, const float* h0, float* out, float* ruc, int B, int T, hipStream_t s);
// Backward recurrence.  wgT/wcT: h-parts transposed: wghT (256,128) = Wg[128:,:]^T, wchT (128,128) = Wc[128:,:]^T.
// dxg (B,T,768) receives pre-activation gradients [dgates | dcand] per direction.  rh (B,T,256) receives r*h_prev
// per direction (for the candidate weight gradient).
struct BiGruBwdWeights {
  const float* wghT[2];
  const float* wchT[2];
};
// dh0 (2,B,128), nullable: gradient w.r.t. the initial state per direction
int launch_bigru_bwd(const float* dout, const float* out, const float* ruc, const BiGruBwdWeights& w, const float* h0,
                     float* dxg, float* rh, float* dh0, int B, int T, hipStream_t s);

// ---------------------------------------------------------------- decoder.hip
// Columns of the decoder's merged [query | cell_output | pad] projection: a power of two so every cluster width slices it
// into wave-reducible column groups.
inline int dec_out_cols(int r) {
  int n = 512;
  while (n < kAtt + kMel * r) n *= 2;
  return n;
}
// Columns of the backward FAN round's [d cell_output | pad] product (same power-of-two rule).
inline int dec_fan_cols(int r) {
  int n = 128;
  while (n < kMel * r) n *= 2;
  return n;
}
// Composite forward weights (workspace; products of consecutive linear maps, see model.hip build_dec_composites)
struct DecComposite {
  const float* wx;     // (128+80r+256, 256)   x = [p2 ; out ; ctx] wx + in_b   (the kernel uses rows [0, 128+80r); the ctx rows feed VWx)
  const float* wg0;    // (128+80r+256+256, 512)  gates_0 pre-activation = [p2 ; out ; h0 ; ctx] wg0 + bg0   (ctx rows feed VWg)
  const float* bg0;    // (512)
  const float* wo;     // (256, NO)  [q | out | 0] = (x + h3) wo + bo
  const float* bo;     // (NO)
  const float* wp1o;   // (256, 256)  pre_net layer 1 of a step fed by the previous cell_output, straight from (x + h3)
  const float* bp1o;   // (256)
  int NO;
};
struct DecWeights {
  const float *pre_w1, *pre_b1, *pre_w2, *pre_b2;  // (80,256) (256,128)
  const float *in_w, *in_b;                        // (384,256)
  const float *gw[3], *gb[3], *cw[3], *cb[3];      // (512,512) (512) (512,256) (256)
  const float *out_w, *out_b;                      // (256,80r)
  const float* q_w;                                // (80r,256)
  const float* att_v;                              // (256)
  const float* att_w;                              // (80r+256,256)  host side only (composites); the kernels read DecComposite
};
// per-(b,t) forward stash record (floats)
constexpr int kStP1 = 0;                  // 256  pre-net layer 1 (post relu, post dropout)
constexpr int kStP2 = 256;                // 128
constexpr int kStX = 384;                 // 256  in-proj output
constexpr int kStH = 640;                 // 3*256 new GRU states
constexpr int kStR = 1408;                // 3*256
constexpr int kStU = 2176;                // 3*256
constexpr int kStC = 2944;                // 3*256
constexpr int kStRH = 3712;               // 3*256 r*h_prev
constexpr int kStCtx = 4480;              // 256
constexpr int kStAtt = 4736;              // 256  (slot kept for layout stability; the attention vector is no longer formed)
constexpr int kStQ = 4992;                // 256
constexpr int kStY = 5248;                // 256  x + h3 (out-proj input)
constexpr int kStRec = 5504;
// per-(b,t) gradient stash record (floats): pre-activation gradients feeding the weight-gradient GEMMs
constexpr int kGsG = 0;                   // 3*512 gates pre-act grads
constexpr int kGsC = 1536;                // 3*256 candidate pre-act grads
constexpr int kGsX = 2304;                // 256 dx (in-proj output grad)
constexpr int kGsQ = 2560;                // 256 dq
constexpr int kGsP1S = 2816;              // 256 d prenet-1 pre-act of step t+1 if that step was fed cell_output[t], else 0
constexpr int kGsCtx = 3072;              // 256 d context
constexpr int kGsP2 = 3328;               // 128 d prenet-2 pre-act
constexpr int kGsP1 = 3456;               // 256 d prenet-1 pre-act
constexpr int kGsO = 3712;                // 80r (<= 400) d cell_output (total)
constexpr int kGsRec = 4112;

struct DecFwdArgs {
  DecWeights w;
  DecComposite c;
  const float* keys;     // (B,Tt,256)
  const float* values;   // (B,Tt,256)
  const float* vwx;      // (B,Tt,256)  values . Wx_c        (context folded into the input projection, see decoder.hip)
  const float* vwg;      // (B,Tt,512)  values . Wx_c Wg0_x  (... and into GRU-1's gates)
  const int32_t* text_length;
  const float* mel;      // (B,Td,80r) or null (inference)
  const uint8_t* keep1;  // (B,Td,256) or null
  const uint8_t* keep2;  // (B,Td,128) or null
  const uint8_t* sample; // (Td,B) or null
  float* out;            // (B,Td,80r)
  float* align;          // (B,Td,Tt)
  float* stash;          // (B,Td,kStRec) or null
  float* prein;          // (B,Td,80) pre-net input frames actually used (train stash) or null
  const float* pre2;     // (B,Td,128) with row pitch ldpre2: pre-net output of every TEACHER-FORCED step, formed before the launch
  int ldpre2;            //   (train: the P2 slot of the stash, which the kernel overwrites on the steps fed by the previous output); null: all in-kernel
  void* xchg;            // decoder_xchg_bytes(B,Tt): granule area for the in-launch all-gathers
  int* err;              // set to 1 by the kernel if a bounded spin timed out
  long long* trace;      // optional (TACO_DEC_TRACE=1): per-phase wall_clock64 stamps of block 0 at step Td/2
  int B, Tt, Td, r;
  int P;                 // cluster width (workgroups per row); chosen by launch_decoder_fwd
  int fakew = 0;         // timing probe (TACO_DEC_FAKEW): see Xchg::fake
  int lres0 = 0, lres1 = 0;  // launch-resident weight rows (LDS) of the GRU-2 / GRU-3 gate mat-vecs; chosen by launch_decoder_fwd
  int xchg_zeroed = 0;       // 1: the caller's batched init launch has zeroed the exchange area already (decoder3.hip only)
  int xcc_table_ofs = 0;     // decoder3.hip: int offset, inside the exchange area, of the 256-entry placement table
  int fast_ok = 1;           // decoder3.hip: 0 forces the placement-independent (agent-scope) publish form (TACO_DEC_V3_AGENT=1)
  int row0 = 0;              // decoder3.hip: first batch row of this launch (B > 32 runs as consecutive launches of <= 32 rows, round 6)
};
int64_t decoder_xchg_bytes(int B, int Tt);
int decoder_last_cluster(int which);   // cluster width (workgroups per row) of the last forward (0) / backward (1) launch
int launch_decoder_fwd(DecFwdArgs a, hipStream_t s);
// decoder3.hip: clusters of 32 workgroups x up to 4 rows with register-resident weights.  TACO_ENOTFOUND (nothing enqueued) when
// the shape is outside its scope (Tt > 256, r not in {2, 5}) or TACO_DEC_V3=0: the caller then takes launch_decoder_fwd.  B > 32 (round 6):
// consecutive launches of <= 32 rows each (DecFwdArgs::row0), i.e. the step cost per 32 rows stays what it is at B = 32.
int launch_decoder3_fwd(DecFwdArgs a, hipStream_t s);
void decoder_note_cluster(int which, int P);

struct DecBwdArgs {
  DecWeights wT;         // every matrix TRANSPOSED (out,in); biases unused
  const float* wot;      // (80r+512, 256)  [Wo^T ; (Wo Wq)^T ; (Wo[:, last frame] W1)^T]
  const float* wdx;      // (256, 256)  Wx_o^T Wo^T: dx_{t+1} -> d(x + h3)_t through cell_output_t
  const float* att_v;    // (256)
  const float* keys;
  const float* vwx;      // (B,Tt,256)  values . Wx_c:  d alignments_t[s] = vwx[s] . dx_{t+1}
  const int32_t* text_length;
  const uint8_t* keep1;
  const uint8_t* keep2;
  const uint8_t* sample;
  const float* dout;     // (B,Td,80r) dLoss/d seq2seq_output (direct: L1 sign + post-net path)
  const float* out;      // (B,Td,80r)
  const float* align;    // (B,Td,Tt)
  const float* stash;    // (B,Td,kStRec)
  float* gstash;         // (B,Td,kGsRec)
  float* dkeys;          // (B,Tt,256) with row pitch ldk, zero-initialised, accumulated
  int ldk;
  float* datt_v;         // (B,256) per-row partial gradients of attention_v (written, not accumulated)
  void* xchg;
  int* err;
  long long* trace;      // optional per-phase stamps (see DecFwdArgs)
  int B, Tt, Td, r;
  int P;
  int fakew = 0;
  int lres0 = 0, lres1 = 0;  // launch-resident weight rows (LDS) of the GRU-3 / GRU-2 gate mat-vecs; chosen by launch_decoder_bwd
  int xcc_table_ofs = 0, fast_ok = 1, row0 = 0;   // decoder3.hip (see DecFwdArgs)
  int xchg_zeroed = 0;       // 1: the caller has zeroed the exchange area on this stream already (taco_backward's batched init launch)
  int hoisted = 0;           // 1: the pre-net gradients of the teacher-forced steps are formed after the launch (model.hip); the kernel
                             //    runs the pre-net backward only where a step was fed by the previous output
};
int launch_decoder_bwd(DecBwdArgs a, hipStream_t s);
int launch_decoder3_bwd(DecBwdArgs a, hipStream_t s);   // decoder3.hip; TACO_ENOTFOUND outside its scope

// ---------------------------------------------------------------- vocoder.hip
// Griffin-Lim (audio.py:77-97).  mag_t / phase0 (B, 1025, F); wave (B, 300 (F - 1)); work: griffinlim_workspace_floats floats
int64_t griffinlim_workspace_floats(int B, int F);
int launch_griffinlim(const float* mag_t, const float* phase0, float* wave, float* work, int B, int F, int n_iter, hipStream_t s);
