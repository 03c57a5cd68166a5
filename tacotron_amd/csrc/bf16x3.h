// bf16x3.h -- fp32 products on the bf16 matrix pipe: exact three-way operand split + six plane products (shared by gemm.hip / gemm2.hip)
#pragma once
#include "common.h"

namespace {

// ---- fp32 products on the bf16 matrix pipe (round 5) ------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at 64 cycles per SIMD for 4,096 FLOP; v_mfma_f32_32x32x16_bf16 at 32 cycles for 32,768.  An
// fp32 value is EXACTLY the sum of three bf16 values (x = h + m + l: h = the top 16 bits of x, m = the top 16 bits of the
// exact remainder x - h, l = (x - h) - m, which has at most 8 significant bits left), so a product a * b is the sum of nine
// exact bf16 x bf16 products, of which the six largest (h h, h m, m h, m m, h l, l h) carry everything above 2^-24 of the
// result -- the size of ONE fp32 rounding -- and the accumulation is fp32 in both forms.  Six bf16 MFMAs replace eight fp32
// MFMAs per 16 k-columns at 1/2 the cycles each: 3/8 of the matrix-pipe time.  Measured against fp64 the six-term form is as
// accurate as the fp32 instruction sequence it replaces (tests/test_gpu_ops.py::test_bf16x3_products_are_fp32_grade; numpy
// model of both: 5.0e-7 vs 5.7e-7 rel-L2 at K = 2048).  The operands stay fp32 in HBM and LDS; the split happens in registers.
// Switch: TACO_GEMM2_BF16X=0 restores v_mfma_f32_32x32x2_f32 everywhere (read on every launch).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Pl3 {
  u32x4 h, m, l;   // 8 bf16 each: element q of the lane's 8 k-slots in bits [16 (q & 1), +16) of word q >> 1
};
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);   // {hi16(x1), hi16(x0)}
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);   // exact
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);   // exact, <= 8 bits
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
__device__ __forceinline__ Pl3 split8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7) {
#ifdef GEMM2_LAB_NOSPLIT   // timing lab: no VALU work at all (results are garbage)
  Pl3 g;
  g.h = u32x4{__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)};
  g.m = u32x4{__float_as_uint(f4), __float_as_uint(f5), __float_as_uint(f6), __float_as_uint(f7)};
  g.l = g.h;
  return g;
#endif
  unsigned h[4], m[4], l[4];
  split2(f0, f1, h[0], m[0], l[0]);
  split2(f2, f3, h[1], m[1], l[1]);
  split2(f4, f5, h[2], m[2], l[2]);
  split2(f6, f7, h[3], m[3], l[3]);
  Pl3 p;
  p.h = u32x4{h[0], h[1], h[2], h[3]};
  p.m = u32x4{m[0], m[1], m[2], m[3]};
  p.l = u32x4{l[0], l[1], l[2], l[3]};
  return p;
}
__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += a * b over the lanes' 16 k-slots, smallest terms first
__device__ __forceinline__ void mfma6(f32x16& acc, const Pl3& a, const Pl3& b) {
#ifdef GEMM2_LAB_NOMFMA    // timing lab: the planes are formed and dropped (results are garbage)
  asm volatile("" ::"v"(a.h), "v"(a.m), "v"(a.l), "v"(b.h), "v"(b.m), "v"(b.l));
  return;
#endif
  acc = mfma_bf(a.l, b.h, acc);
  acc = mfma_bf(a.h, b.l, acc);
  acc = mfma_bf(a.m, b.m, acc);
  acc = mfma_bf(a.m, b.h, acc);
  acc = mfma_bf(a.h, b.m, acc);
  acc = mfma_bf(a.h, b.h, acc);
}
// Two-accumulator form (round 6, the default of both kernels; -DTACO_BF16X_ACC1 = one accumulator, rounds 5 / early 6): the five
// low-order plane products go to `lo`, h h to `hi`; the kernels add the two once behind their k-loop.  The matrix pipe rounds every
// product onto the grid of the accumulator it is added to (bf16x_max_chain below): against `lo` -- ~2^-7 of `hi` -- the l h / h l
// products (2^-16 of a product) keep ~20 more bits than against the full sum, and `hi` only ever receives 16-bit products.
// Measured (profiles/r06_bf16x3_acc2.txt): mixed-sign operands 0.3-0.4 x the error of the fp32 INSTRUCTION at every depth (one
// accumulator: 0.9-1.2 x); same-signed full mantissas 1.4e-7 / 1.8e-7 / 3.2e-7 at chains of 256 / 1024 / 2048 (3.8e-7 / 8.5e-7 /
// 1.9e-6), 2.4e-6 at 6144 (1.7e-5); cost: 64 accumulator registers (the NN kernel: 229 -> 242 VGPRs, still two workgroups per CU),
// +0.5-1 % kernel time.
__device__ __forceinline__ void mfma6_2(f32x16& hi, f32x16& lo, const Pl3& a, const Pl3& b) {
  lo = mfma_bf(a.l, b.h, lo);
  hi = mfma_bf(a.h, b.h, hi);
  lo = mfma_bf(a.h, b.l, lo);
  lo = mfma_bf(a.m, b.m, lo);
  lo = mfma_bf(a.m, b.h, lo);
  lo = mfma_bf(a.h, b.m, lo);
}
__device__ __forceinline__ Pl3 zero_pl3() {
  Pl3 p;
  p.h = u32x4{0, 0, 0, 0}; p.m = p.h; p.l = p.h;
  return p;
}


// TACO_GEMM2_BF16X (read on every launch): 1 (default) = the fp32 products of the big GEMM kernels are formed on the bf16 matrix
// pipe from exact three-way operand splits (fp32-grade results, 3/8 of the matrix-pipe time); 0 = v_mfma_f32_32x32x2_f32, the
// form of rounds 2-4 (A/B runs, bisecting).
inline bool env_bf16x() {
  const char* e = getenv("TACO_GEMM2_BF16X");
  return !(e && atoi(e) == 0);
}

// Longest accumulation CHAIN (k-products added into one accumulator register) a launch may have on the bf16x3 form; longer chains
// take the fp32 MFMA form.  Why (round 6; tools/micro/mfma_bf16_probe.hip, tools/bf16x3_chain_probe.py, profiles/r06_mfma_probe.txt,
// r06_bf16x3_chain.txt, r06_bf16x3_acc2.txt): the matrix pipe rounds every PRODUCT by itself onto the accumulator's grid (2^-26 of
// its leading bit; both instructions do: 16 products of 1/16 ulp(C) each vanish although their sum is a whole ulp) before adding.
// For the fp32 instruction that is an eighth of an ulp per product.  For the split form it means that low-order plane products
// (l h, h l: 2^-16 of a product) added into the FULL sum drop out entirely once that sum exceeds ~2^11 products' worth -- invisible
// with mixed signs, one-sided with same-signed operands (one accumulator, all-ones mantissas: 1.9e-6 at a chain of 2048, 1.7e-5 at
// 6144; fp32 form 2-5e-8).  The second accumulator (mfma6_2) removes most of it -- 3.2e-7 at 2048, i.e. inside a 4e-7 bar -- but at
// 4096 / 6144 the same-signed worst case is still 1.3e-6 / 2.4e-6, so beyond 2048 the bf16x3 form is not used.  Every launch of the
// train step is inside the bound (deepest: encoder conv bank width 16 = 16 x 128 = 2048; the deeper reductions -- proj1, the bank
// input gradient -- are k-split into chains of 768-1741, the weight gradients into row ranges of 320-720); what the bound catches is
// a plain deep call (taco_conv_gemm with K x taps > 2048) and TACO_DETERMINISTIC=1's one-workgroup-per-tile weight gradients.
// TACO_BF16X_MAX_CHAIN overrides (probing).
inline int bf16x_max_chain() {
  const char* e = getenv("TACO_BF16X_MAX_CHAIN");
  return e ? atoi(e) : 2048;
}

}  // namespace
