// What does v_mfma_f32_32x32x16_bf16 do with products that are small against its fp32 accumulator?  (Round 6: the adversarial
// bf16x3 test -- same-signed operands with full mantissas, K = 6144 -- measured 1.7e-5 relative error where the fp32 MFMA form
// has 5e-8.)  One wave, one instruction per case; A = all `a`, B = all `b`, C = all `c`, so every element of D is
// c + 16 a b in exact arithmetic.  Cases scan the ratio between the products and the accumulator and the sign, to tell apart
//   (i)  exact sum of the 16 products, ONE rounding (to nearest) with C                (D - C follows 16 a b until it is < ulp / 2)
//   (ii) products aligned to C and truncated one by one                               (a product below some fraction of ulp(C) vanishes)
//   (iii) round-toward-zero / toward -inf of the final sum                            (positive and negative cases differ)
// build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_bf16_probe tools/micro/mfma_bf16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_bf16(const unsigned* ab, const float* c, float* d, int chain) {
  // ab[0] = packed pair of bf16 `a`, ab[1] = packed pair of bf16 `b` (same value in both halves unless the case says otherwise),
  // ab[2..3]: optional second (a, b) pair used by k-slots 4..7 of every lane (mixed-magnitude cases)
  u32x4 A = {ab[0], ab[0], ab[2], ab[2]}, B = {ab[1], ab[1], ab[3], ab[3]};
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = c[0];
  for (int i = 0; i < chain; ++i)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
  if (threadIdx.x == 0) d[0] = acc[0];
  if (threadIdx.x == 37) d[1] = acc[5];
}
__global__ void k_f32(const float* ab, const float* c, float* d, int chain) {
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = c[0];
  for (int i = 0; i < chain; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[0], ab[1], acc, 0, 0, 0);
  if (threadIdx.x == 0) d[0] = acc[0];
  if (threadIdx.x == 37) d[1] = acc[5];
}
static unsigned bf16_pair(float x) {   // x must be bf16-exact
  unsigned u;
  memcpy(&u, &x, 4);
  return (u >> 16) | (u & 0xffff0000u);
}
int main() {
  unsigned* dab; float *dc, *dd, *dabf;
  hipMalloc(&dab, 16); hipMalloc(&dabf, 8); hipMalloc(&dc, 4); hipMalloc(&dd, 8);
  auto run = [&](float a, float b, float a2, float b2, float c, int chain) {
    unsigned h[4] = {bf16_pair(a), bf16_pair(b), bf16_pair(a2), bf16_pair(b2)};
    hipMemcpy(dab, h, 16, hipMemcpyHostToDevice);
    hipMemcpy(dc, &c, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, dab, dc, dd, chain);
    float r[2];
    hipMemcpy(r, dd, 8, hipMemcpyDeviceToHost);
    return r[0] == r[1] ? r[0] : NAN;
  };
  auto runf = [&](float a, float b, float c, int chain) {
    float h[2] = {a, b};
    hipMemcpy(dabf, h, 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, &c, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dabf, dc, dd, chain);
    float r[2];
    hipMemcpy(r, dd, 8, hipMemcpyDeviceToHost);
    return r[0];
  };
  const float ulp1 = ldexpf(1.f, -23);   // ulp of C = 1
  printf("== A: C = 1, 16 equal products p = 2^-s each (exact sum 16 p = 2^(4-s)); D - C in ulps of C, for +p and -p\n");
  for (int s = 18; s <= 34; ++s) {
    const float p = ldexpf(1.f, -s);
    const float dp = run(1.f, p, 1.f, p, 1.f, 1) - 1.f, dm = run(1.f, -p, 1.f, -p, 1.f, 1) - 1.f;
    printf("  s=%2d  p = %7.4f ulp  exact 16p = %8.4f ulp   D-C(+) = %8.4f ulp   D-C(-) = %8.4f ulp\n", s, p / ulp1, 16 * p / ulp1, dp / ulp1, dm / ulp1);
  }
  printf("== B: C = 1.5 (odd last bit patterns), 16 products of 3/64 ulp .. : rounding mode of the final sum\n");
  for (int num = 1; num <= 15; num += 2) {   // 16 p = num/8 ulp ... choose p = num * 2^-7 ulp -> 16 p = num / 8 ulp
    const float p = num * ldexpf(1.f, -30);  // = num * 2^-7 ulp1
    for (float c : {1.f, 1.f + ulp1}) {
      const float dp = run(1.f, p, 1.f, p, c, 1) - c, dm = run(1.f, -p, 1.f, -p, c, 1) - c;
      printf("  C=%s 16p = %5.3f ulp   D-C(+) = %5.2f ulp   D-C(-) = %5.2f ulp\n", c == 1.f ? "1     " : "1+ulp ", 16 * p / ulp1, dp / ulp1, dm / ulp1);
    }
  }
  printf("== C: mixed magnitudes inside one instruction: 8 products of 1 (k-slots 0-3 of both lane halves) + 8 products of 2^-s; C = 0 and C = 1\n");
  for (int s = 20; s <= 30; s += 1) {
    const float p = ldexpf(1.f, -s);
    const float d0 = run(1.f, 1.f, 1.f, p, 0.f, 1), d1 = run(1.f, 1.f, 1.f, p, 1.f, 1);
    printf("  s=%2d  exact = 8 + %g   D(C=0) - 8 = %g (%.3f of exact)   D(C=1) - 9 = %g\n", s, 8 * p, d0 - 8.f, (d0 - 8.f) / (8 * p), d1 - 9.f);
  }
  printf("== D: chains: C = 1, then n x (16 products of p = 2^-27 = 1/16 ulp -> exactly +1 ulp per instruction)\n");
  for (int n : {1, 4, 16, 64, 256}) {
    const float p = ldexpf(1.f, -27);
    printf("  n=%3d  D-C = %7.2f ulp (exact %d)   with -p: %7.2f\n", n, (run(1.f, p, 1.f, p, 1.f, n) - 1.f) / ulp1, n, (run(1.f, -p, 1.f, -p, 1.f, n) - 1.f) / ulp1);
  }
  for (int n : {1, 4, 16, 64, 256}) {
    const float p = ldexpf(1.f, -28);  // 16 p = 0.5 ulp per instruction
    printf("  n=%3d  16p = 0.5 ulp per instruction: D-C = %7.2f ulp (exact %.1f)   with -p: %7.2f\n", n, (run(1.f, p, 1.f, p, 1.f, n) - 1.f) / ulp1, n * 0.5,
           (run(1.f, -p, 1.f, -p, 1.f, n) - 1.f) / ulp1);
  }
  for (int n : {1, 4, 16, 64, 256}) {
    const float p = ldexpf(1.f, -29);  // 16 p = 0.25 ulp per instruction
    printf("  n=%3d  16p = 0.25 ulp per instruction: D-C = %7.2f ulp (exact %.2f)   with -p: %7.2f\n", n, (run(1.f, p, 1.f, p, 1.f, n) - 1.f) / ulp1, n * 0.25,
           (run(1.f, -p, 1.f, -p, 1.f, n) - 1.f) / ulp1);
  }
  printf("== E: the fp32 instruction (v_mfma_f32_32x32x2_f32: 2 products per instruction), same questions\n");
  for (int s = 22; s <= 27; ++s) {
    const float p = ldexpf(1.f, -s);
    printf("  s=%2d  exact 2p = %6.3f ulp   D-C(+) = %6.3f ulp   D-C(-) = %6.3f ulp\n", s, 2 * p / ulp1, (runf(1.f, p, 1.f, 1) - 1.f) / ulp1, (runf(1.f, -p, 1.f, 1) - 1.f) / ulp1);
  }
  for (int n : {16, 256}) {
    const float p = ldexpf(1.f, -25);  // 2p = 0.5 ulp
    printf("  chain n=%3d, 2p = 0.5 ulp per instruction: D-C = %7.2f ulp (exact %.1f)   with -p: %7.2f\n", n, (runf(1.f, p, 1.f, n) - 1.f) / ulp1, n * 0.5,
           (runf(1.f, -p, 1.f, n) - 1.f) / ulp1);
    const float q = ldexpf(1.f, -26);  // 2q = 0.25 ulp
    printf("  chain n=%3d, 2p = 0.25 ulp per instruction: D-C = %7.2f ulp (exact %.2f)   with -p: %7.2f\n", n, (runf(1.f, q, 1.f, n) - 1.f) / ulp1, n * 0.25,
           (runf(1.f, -q, 1.f, n) - 1.f) / ulp1);
  }
  return 0;
}
