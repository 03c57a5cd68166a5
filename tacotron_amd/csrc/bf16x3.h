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

}  // namespace
