/* taco_hip.h -- C ABI of libtaco_hip.so, the MI355X (gfx950) implementation of the Tacotron acoustic-model
 * hot path of barronalex/Tacotron (models/tacotron.py + models/ops.py).
 *
 * The reference has NO FFI / plugin boundary (it is a pure-Python TF-1.2 graph), so every entry
 * point below cites the reference function whose arithmetic it replaces.  Conventions:
 *   - every function returns int: 0 = ok, negative = TACO_E* (never throws, never exits);
 *     taco_last_error_string() describes the last failure on the calling thread;
 *   - all pointers except `shape`/tables are DEVICE pointers, contiguous row-major, fp32 unless stated;
 *   - one call = stream-ordered enqueues on `stream` (a hipStream_t passed as void*); no allocation, no
 *     host synchronisation, no ownership transfer.  The caller supplies parameters, inputs, outputs and a
 *     workspace of taco_workspace_bytes() bytes;
 *   - tensors are batch-major (B, T, C); weights use the reference's TF variable layouts
 *     (dense kernel (in,out); conv1d kernel (k,Cin,Cout); GRUCell gates kernel (Cin+H,2H) r-then-u ...).
 */
#ifndef TACO_HIP_H
#define TACO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define TACO_VERSION 118

#define TACO_OK 0
#define TACO_EINVAL (-1)   /* bad argument / unsupported shape   */
#define TACO_ELAUNCH (-2)  /* HIP launch or runtime error        */
#define TACO_ENOTFOUND (-3)

/* activation codes for taco_conv_gemm */
#define TACO_ACT_NONE 0
#define TACO_ACT_RELU 1
#define TACO_ACT_SIGMOID 2
#define TACO_ACT_TANH 3

/* Problem shape.  Model widths (embed 256, prenet 256/128, CBHG 128, attention/decoder 256, 80 mels, 1025 bins,
 * K=16/8 conv banks) are the reference's Config constants (tacotron.py:12-33, 131, 147) and are compiled in. */
typedef struct TacoShape {
  int32_t B;   /* batch (Config.batch_size = 32)                                    */
  int32_t Tt;  /* padded text length                                               */
  int32_t Td;  /* decoder steps = Config.max_decode_iter (tacotron.py:13)          */
  int32_t r;   /* mel frames per decoder step (audio.r, audio.py:15-17)            */
  int32_t V;   /* vocab_size (train.py:22)                                         */
  int32_t S;   /* num_speakers (tacotron.py:23); <= 1 = single speaker, no speaker path */
} TacoShape;

/* One row of the parameter / workspace tables. */
typedef struct TacoTensorInfo {
  char name[64];
  int64_t offset; /* in floats from the base pointer */
  int64_t size;   /* in floats                      */
  int32_t ndim;
  int32_t dims[4];
} TacoTensorInfo;

int taco_version(void);
const char* taco_last_error_string(void);

/* ---- parameter layout: one flat fp32 buffer, TF variable order (tacotron.py:107-154 graph order) ---------- */
int64_t taco_param_count(const TacoShape* shape);
/* Fills up to `cap` rows; returns the number of tensors (or negative error). */
int taco_param_table(const TacoShape* shape, TacoTensorInfo* rows, int cap);

/* ---- workspace ------------------------------------------------------------------------------------------ */
/* Bytes needed by taco_forward / taco_backward / taco_infer for `shape` (train != 0 includes backward stashes). */
int64_t taco_workspace_bytes(const TacoShape* shape, int train);
/* Named intermediate tensors inside the workspace (for parity tests / debugging). */
int taco_workspace_table(const TacoShape* shape, int train, TacoTensorInfo* rows, int cap);

/* ---- op level ------------------------------------------------------------------------------------------- */
/* C[m, n] = post( act( sum_{tap<taps} sum_{k<K} A[row(m,tap), k] * W[tap][k][n] + bias[n] ) )
 *   row(m,tap): m = b*T + t  ->  t' = t + tap - pad_l ; zero row unless 0 <= t' < T   ('same' conv1d, ops.py:54-60;
 *   taps = 1, pad_l = 0 is tf.layers.dense, tacotron.py:40-43).
 *   post(y) = (keep ? y * keep[m,n] * 2 : y) * scale[n] + shift[n] + residual[m,n]   (each optional / nullable)
 *   Cpre (nullable) receives the value before scale/shift/residual.  W tap stride is K*ldw floats. */
int taco_conv_gemm(const float* A, int lda, const float* W, int ldw, const float* bias, const float* scale,
                   const float* shift, const float* residual, int ldr, const uint8_t* keep, float* C, int ldc,
                   float* Cpre, int M, int N, int K, int taps, int T, int pad_l, int act, void* stream);

/* Same contract (without the post() part), run the way the library runs the tall-skinny CBHG projections: the (tap, k) sum is
 * cut into chunks that become independent workgroups writing partial slabs (scratch `slabs`, `slab_floats` floats), summed in a
 * fixed order by a second pass -- deterministic, no atomics.  Exposed for parity tests and tuning. */
/* debug: only the eligible NN launches whose running index falls in [lo, hi) use gemm2.hip's kernel (bisecting a divergence);
 * returns the number of eligible (non-pooled) launches seen since the previous call and restarts the count */
int taco_debug_gemm2_window(int lo, int hi);
/* Pre-split weight images (round 6; csrc/kernels.h "weight images"): the bf16 plane image of a weight tensor W (taps, K, N; row pitch
 * ldw) that gemm2.hip's NN kernel reads instead of splitting W in registers.  The model-level entry points build the images of
 * their own weights themselves (into the workspace); this is the op-level door for parity tests and tools.
 *   W == NULL: clears this host thread's table, returns the number of launches that ran the image form since the previous such
 *   call (all threads).   img == NULL: returns the bytes image(W) needs.
 *   otherwise: registers image(W) at img (16-byte aligned, img_bytes >= that size) and enqueues its build on `stream`; the
 *   following taco_conv_gemm / taco_debug_conv_gemm_* calls of this thread whose weight pointer, pitch, taps, K equal W's and whose
 *   N <= N run the image form.  Replaces nothing in the reference (TF holds fp32 kernels only; models/ops.py:54-60,80-86). */
int64_t taco_debug_weight_image(const float* W, int ldw, int taps, int K, int N, void* img, int64_t img_bytes, void* stream);
/* debug / test aid: dense layer with weight rows zero-padded to nld loadable columns (the final 256 -> 1025 layer's form) */
int taco_debug_conv_gemm_nld(const float* A, int lda, const float* W, int ldw, int nld, const float* bias, float* C, int ldc, int M,
                             int N, int K, int act, void* stream);
int taco_debug_conv_gemm_ksplit(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                                int N, int K, int taps, int T, int pad_l, int act, float* slabs, int64_t slab_floats,
                                void* stream);

/* dW[tap][k][n] (+)= sum_m A[row(m,tap), k] * dY[m, n]   (weight gradient of the op above; accumulate != 0 adds) */
int taco_gemm_tn(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, int M, int N, int K, int taps,
                 int T, int pad_l, int accumulate, void* stream);

/* Reference GEMM on scalar FMAs (debug aid for the MFMA kernels; same contract as taco_conv_gemm without post). */
int taco_debug_gemm_naive(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                          int M, int N, int K, int taps, int T, int pad_l, int act, void* stream);

/* Bidirectional GRU(128) over the full padded length (ops.py:117-128; tf.nn.bidirectional_dynamic_rnn without
 * sequence_length).  x (B,T,128); weights in TF GRUCell layout: wg (256,256), bg (256), wc (256,128), bc (128) per
 * direction.  out (B,T,256) = concat(fw,bw).  ruc (B,T,768) receives r,u,c for both directions (nullable).
 * xg (B,T,768) is scratch for the hoisted input projections. */
int taco_bigru_fwd(const float* x, const float* wg_fw, const float* bg_fw, const float* wc_fw, const float* bc_fw,
                   const float* wg_bw, const float* bg_bw, const float* wc_bw, const float* bc_bw, float* xg,
                   float* out, float* ruc, int B, int T, void* stream);

/* ---- model level ---------------------------------------------------------------------------------------- */
/* Tacotron.inference with train=True (tacotron.py:107-154) + add_loss_op (tacotron.py:156-165).
 *   text (B,Tt) int32; text_length (B) int32; mel (B,Td,80r); stft (B,Td,1025r);
 *   speaker (B) int32 speaker ids, used (and required) only when shape->S > 1 (tacotron.py:117-124, ops.py:101-115);
 *   masks (uint8 0/1, nullable = no dropout / no sampling):
 *     enc_keep1 (B,Tt,256), enc_keep2 (B,Tt,128)  encoder pre_net dropout keep masks (tacotron.py:128)
 *     dec_keep1 (B,Td,256), dec_keep2 (B,Td,128)  decoder pre_net dropout keep masks (tacotron.py:64-71)
 *     sample (Td,B): 1 => step t+1 of row b is fed cell_output[t] (ScheduledOutputTrainingHelper, tacotron.py:84-85)
 *   outputs: seq2seq_output (B,Td,80r), output (B,Td,1025r), alignments (B,Td,Tt), loss[3] = {total, seq2seq, output}. */
int taco_forward(const TacoShape* shape, const float* params, const int32_t* text, const int32_t* text_length,
                 const int32_t* speaker, const float* mel, const float* stft, const uint8_t* enc_keep1, const uint8_t* enc_keep2,
                 const uint8_t* dec_keep1, const uint8_t* dec_keep2, const uint8_t* sample, float* seq2seq_output,
                 float* output, float* alignments, float* loss, void* workspace, void* stream);

/* Gradient of loss w.r.t. every parameter (opt.compute_gradients, tacotron.py:172), after taco_forward on the same
 * workspace with the same PARAMETERS, inputs and masks (taco_forward also leaves the transposed weight copies the
 * backward pass reads in the workspace).  seq2seq_output and alignments are the tensors taco_forward produced.
 * grads has taco_param_count floats and is overwritten. */
int taco_backward(const TacoShape* shape, const float* params, const int32_t* text, const int32_t* text_length,
                  const int32_t* speaker, const float* seq2seq_output, const float* alignments, const uint8_t* enc_keep1,
                  const uint8_t* enc_keep2, const uint8_t* dec_keep1, const uint8_t* dec_keep2, const uint8_t* sample,
                  float* grads, void* workspace, void* stream);

/* Tacotron.inference with train=False (test.py:29, ops.InferenceHelper ops.py:5-25): zeros first frame, feeds back
 * its own output, always Td steps, no dropout.  mel/stft/loss absent. */
int taco_infer(const TacoShape* shape, const float* params, const int32_t* text, const int32_t* text_length,
               const int32_t* speaker,
               float* seq2seq_output, float* output, float* alignments, void* workspace, void* stream);

/* add_train_op (tacotron.py:167-185): global-norm clip (cap_grads) then TF-form Adam, in place.
 *   step = global_step after this update (1-based).  scratch: >= 256 floats.  gnorm_out[0] receives ||g||. */
int taco_clip_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float cap,
                        int64_t step, float* scratch, float* gnorm_out, void* stream);

/* Same, guarded: err_words (nullable) points at the two int32 decoder error words of the workspace (tensor "dec.err" of
 * taco_workspace_table: [0] forward, [1] backward).  The persistent decoder kernels exchange data between workgroups with
 * bounded spins; a time-out (workgroups not co-resident on a busy GPU) sets a word and the launch drains with garbage
 * gradients.  When either word is non-zero the update is skipped and gnorm_out[0] = -1.  The words are STICKY: only
 * taco_clear_error resets them (call it once on a fresh workspace, and after handling an error).
 * scratch: >= 256 floats (per-block partial sums of squares, summed in a fixed order: the norm is reproducible). */
int taco_clip_adam_step_guarded(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float cap,
                                int64_t step, float* scratch, float* gnorm_out, const int32_t* err_words, void* stream);
int taco_clear_error(const TacoShape* shape, int train, void* workspace, void* stream);

/* Process-wide decoder mode.  The default persistent decoder kernels (decoder3.hip) exchange data between the 32 workgroups of
 * a cluster through granules that are published with WORKGROUP-scope stores and read with L1-bypassing agent-scope loads when
 * the whole cluster sits on one XCD (verified at kernel start from HW_REG_XCC_ID).  That is faster than the placement-
 * independent agent-scope form (0.21 vs 0.37 us per hop) but leans on gfx942 / gfx950 implementation behaviour: a write-through
 * L1 and one L2 per XCD shared by its CUs.  Granules are epoch-tagged, so a stale read can only delay or time out, never
 * corrupt; a time-out sets the sticky error word (taco_clip_adam_step_guarded).
 *   mode 0 (default)  decoder3.hip, XCD-local exchange where placement allows
 *   mode 1            decoder3.hip, agent-scope exchange only (environment TACO_DEC_V3_AGENT=1)
 *   mode 2            decoder.hip (round-1/2 kernels: one cluster of 8-16 workgroups per batch row; environment TACO_DEC_V3=0)
 * taco_decoder_mode(mode) sets the mode (mode < 0: query only) and returns the previous one.  The Python host escalates
 * 0 -> 1 -> 2 when Tacotron.check() finds an error word set, so a box where the fast form misbehaves degrades instead of
 * skipping every update. */
int taco_decoder_mode(int mode);

/* ---- data-parallel overlap (no reference counterpart: train.py:24 is a single Session) ---------------- */
/* The flat gradient buffer becomes final in FIVE contiguous segments, in this order during taco_backward:
 *   segment 4 = [bounds[4], bounds[5])  post-net CBHG + final dense   (final before the decoder BPTT; ANNOUNCED right after it)
 *   segment 3 = [bounds[3], bounds[4])  attention memory layer + decoder
 *   segment 2 = [bounds[2], bounds[3])  encoder CBHG without its conv bank: projections, highways, bi-GRU
 *   segment 1 = [bounds[1], bounds[2])  encoder conv bank   (behind its grouped weight-gradient launch, the last big one of the pass)
 *   segment 0 = [bounds[0], bounds[1])  embedding(s) + encoder pre_net   (end of taco_backward)
 * taco_grad_segments fills bounds[6] (float offsets) and returns 5 (version 117 and earlier: four segments, the conv bank in
 * segment 0).  taco_wait_grad_segment makes `stream` wait (device
 * side, hipStreamWaitEvent) until segment `seg` of the most recent taco_backward enqueued by the calling thread on the
 * current device is final, so an all-reduce enqueued on `stream` afterwards overlaps the rest of the backward pass.
 * Segment 4 is announced AFTER the decoder BPTT kernel although it is final before it: that kernel is a persistent launch
 * whose 256 workgroups must all be co-resident (one per CU), so no collective is ever allowed to compete with it for CUs;
 * segments 4 and 3 (13.7 MB) travel under the encoder backward, segment 2 (4.7 MB) under the encoder conv-bank gradients,
 * segment 1 (8.9 MB) under the step's tail (bank input gradient, pre_net chain, embedding scatter: ~0.1 ms), and only
 * segment 0 (0.5 MB) is exposed in full. */
int taco_grad_segments(const TacoShape* shape, int64_t* bounds);
int taco_wait_grad_segment(int seg, void* stream);
/* Communication-kernel stand-in for the co-residency tests (tests/test_gpu_dist.py): `blocks` workgroups x `threads` threads, `lds_bytes` of LDS each, spinning
 * for `usec` microseconds.  Does no work. */
int taco_debug_spin(int blocks, int threads, int lds_bytes, int usec, void* stream);

/* Shader-clock probe: one 512-thread workgroup per CU, every wave a chain of `iters` dependent FMAs; out3[0] = elapsed shader
 * cycles, out3[1] = elapsed ticks of the constant 100 MHz counter of workgroup 0 (device int64[3]).  cycles / (ticks * 10 ns) =
 * the clock the chip sustains under a chip-wide latency-bound load, which is what the persistent decoder / bi-GRU kernels scale
 * with (boxes of one pool were measured ~10 % apart on those kernels). */
int taco_debug_clock_probe(long long* out3, int iters, void* stream);

/* Fabric probe: the latencies the persistent decoder / bi-GRU kernels wait on, measured on the box at hand (bench.py `box`).
 * One launch of 64 small workgroups; out32 (device int64[32], zeroed by the caller) receives, in ticks of the 100 MHz counter,
 * the total of `iters` round trips of a granule ping-pong between two workgroups of ONE XCD with workgroup-scope stores [0]
 * (decoder3.hip's exchange) and agent-scope stores [1], between two XCDs [2] (decoder.hip's exchange), of `iters` dependent loads
 * that hit the L2 [3] / agent-scope loads of cold lines anywhere in the scratch buffer [4], and the time
 * one workgroup needs to stream 8 MB [5]; [6] = iters, [7] = ok bits, [8 + b] = XCC id of workgroup b < 24.
 * gran4k: 4 KiB of zeroed device memory; scratch: zeroed device memory, scratch_bytes >= 32 MiB.  No counterpart in the reference. */
int taco_debug_fabric_probe(long long* out32, void* gran4k, const void* scratch, long long scratch_bytes, int iters, void* stream);

/* ---- spectrogram boundary ---------------------------------------------------------------------------- */
/* test.py:64 `out * stft_std + stft_mean` followed by audio.reshape_frames(forward=False) (audio.py:29-35), on the device.
 *   output (B, Td, r*C) as produced by taco_infer (C = 1025 linear bins, or 80 for mel frames); stft_mean / stft_std (r*C)
 *   spec  (B, F, C), nullable: chronological, de-normalised log-magnitude frames, F = (Td / 4) * 4 * r
 *   mag_t (B, C, F), nullable: exp(spec) transposed -- the matrix audio.invert_spectrogram (audio.py:69-72) gives Griffin-Lim */
int taco_denorm_unframe(const float* output, const float* stft_mean, const float* stft_std, float* spec, float* mag_t,
                        int B, int Td, int r, int C, void* stream);

/* ---- vocoder ---------------------------------------------------------------------------------------------- */
/* audio.griffinlim (audio.py:77-97) with the reference's constants compiled in (n_fft 2048, win_length 1200, hop_length 300,
 * periodic Hann, librosa center=True framing): n_iter rounds of istft -> stft keeping the given magnitudes, then a final istft.
 *   mag_t  (B, 1025, F)  linear magnitudes (taco_denorm_unframe's mag_t)
 *   phase0 (B, 1025, F)  initial phase angles in radians (the reference draws 2 pi U[0,1), audio.py:81; caller supplied here so
 *                        that results are reproducible)
 *   wave   (B, 300 (F - 1)) output samples
 *   workspace: taco_griffinlim_workspace_bytes(B, F) bytes.  Hand-written 2048-point FFT, no vendor library. */
int64_t taco_griffinlim_workspace_bytes(int B, int F);   /* F >= 5 frames (as taco_griffinlim), else TACO_EINVAL */
int taco_griffinlim(const float* mag_t, const float* phase0, float* wave, void* workspace, int B, int F, int n_iter, void* stream);

/* Bernoulli(p_keep) bytes from a counter-based hash RNG (replaces TF's dropout / Bernoulli sampler state). */
int taco_fill_bernoulli(uint8_t* out, int64_t n, float p_one, uint64_t seed, void* stream);

/* ---- measurement --------------------------------------------------------------------------------------------- */
/* Launch-level timing with hipEventRecord pairs on the launch stream (rings of 4096 event pairs per category; no
 * synchronisation at record time).  Categories: 0 = persistent decoder forward kernel, 1 = decoder backward kernel,
 * 2 = MFMA GEMM family (conv_gemm / gemm_tn / fused highway stack launches), 3 = bi-GRU recurrences.
 * taco_profile_enable(mask): bit c switches category c on (mask 0 = off); bit 4 (16) additionally keeps ALL work on the caller's
 * stream (no side stream) while set, so that a category-2 pass times every GEMM launch by itself.
 * taco_profile_read2 synchronises on the recorded events, writes up to `cap` elapsed times (milliseconds) and the
 * algorithmic FLOPs of each launch (2 M N K taps) to the HOST arrays, oldest first, clears the ring, returns the count.
 * Launches that overlap on two streams are each timed by their own events (their times then sum to more than the wall). */
int taco_profile_enable(int mask);
/* Workgroups per cluster the most recent decoder forward (which = 0) / backward (1) launch of this process ran with:
 * 32 on the default path (decoder3.hip: 8 clusters x 32 workgroups, each cluster owning up to 4 batch rows); on the fallback
 * path (decoder.hip: one cluster per batch row) 8 in training and up to 16 at inference, fewer when B * width workgroups
 * would not be co-resident. */
int taco_debug_last_cluster(int which);
/* Labels of the launches currently in ring `which` (what each timed launch was: kernel family and shape), one line per
 * launch, oldest first, into the HOST buffer; returns the number of launches.  Call BEFORE taco_profile_read2 (which clears). */
int taco_debug_profile_labels(int which, char* buf, int cap);
int taco_profile_read(int which, float* ms, int cap);
int taco_profile_read2(int which, float* ms, double* flops, int cap);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* TACO_HIP_H */
